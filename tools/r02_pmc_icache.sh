#!/bin/bash
# instruction-cache counters of the kernels whose name matches $2
O="$GRAFT_REPO_ROOT/gpurun_out/${1:-r02_pmci}"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_IFETCH -d "$O/pmc" -o k --output-format csv -- python3 "$GRAFT_REPO_ROOT/bench.py" --gpus 1 --steps 20 --warmup 2 --no-cpu-baseline --no-reupload-leg > "$O/b.json" 2> "$O/b.err" || { tail -5 "$O/b.err"; exit 1; }
python3 - "$O" "${2:-window}" <<'PY'
import csv,sys,glob,collections
f=glob.glob(sys.argv[1]+"/pmc/**/k_counter_collection.csv", recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"].split("(")[0].replace("void ","")
    if any(k in n for k in sys.argv[2].split(",")): acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n,c in acc.items():
    print(n, {k: round(sum(v)/len(v)) for k,v in c.items()})
PY
