#!/bin/bash
# kernel-trace stats of the driver's command (bench.py --gpus 1 --steps 20 --warmup 5)
O="$GRAFT_REPO_ROOT/gpurun_out/${1:-r02_prof20}"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d "$O/kt" -o k --output-format csv -- python3 "$GRAFT_REPO_ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-reupload-leg > "$O/bench.json" 2> "$O/bench.err"
echo rc=$?
find "$O" -name "*kernel_trace.csv" -delete
python3 - "$O" <<'PY'
import csv,sys,glob
f=glob.glob(sys.argv[1]+"/kt/**/k_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print("%-60s calls=%4s avg=%9.1f us total=%8.2f ms" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
