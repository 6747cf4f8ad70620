#!/bin/bash
# per-kernel average durations of the driver-shaped bench under rocprofv3 (pass extra env via the caller's environment)
O="$GRAFT_REPO_ROOT/gpurun_out/${1:-r02_kstats}"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$O/kt" -o k --output-format csv -- python3 "$GRAFT_REPO_ROOT/bench.py" --gpus 1 --steps ${2:-20} --warmup 5 --no-cpu-baseline --no-reupload-leg > "$O/b.json" 2> "$O/b.err"
find "$O" -name "*kernel_trace.csv" -delete
python3 - "$O" <<'PY'
import csv,sys,glob
f=glob.glob(sys.argv[1]+"/kt/**/k_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r["AverageNs"]) > 8000: print("%-44s calls=%3s avg=%8.1f us" % (r["Name"].split("(")[0].replace("void ","")[:44], r["Calls"], float(r["AverageNs"])/1e3))
PY
