#!/bin/bash
# SQ counters of the kernels whose name matches $2 in the driver-shaped bench
O="$GRAFT_REPO_ROOT/gpurun_out/${1:-r02_pmck}"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU -d "$O/pmc" -o k --output-format csv -- python3 "$GRAFT_REPO_ROOT/bench.py" --gpus 1 --steps 20 --warmup 2 --no-cpu-baseline --no-reupload-leg > "$O/b.json" 2> "$O/b.err"
python3 - "$O" "${2:-pair}" <<'PY'
import csv,sys,glob,collections
f=glob.glob(sys.argv[1]+"/pmc/**/k_counter_collection.csv", recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"].split("(")[0].replace("void ","")
    if sys.argv[2] in n: acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n,c in acc.items():
    print(n, {k: round(sum(v)/len(v)) for k,v in c.items()}, "dispatches", len(next(iter(c.values()))))
PY
