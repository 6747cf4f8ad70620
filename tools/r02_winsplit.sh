#!/bin/bash
O="$GRAFT_REPO_ROOT/gpurun_out/r02_winsplit"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for M in split nosplit; do
  if [ $M = nosplit ]; then export H2V_MSM_WIN_NOSPLIT=1; else unset H2V_MSM_WIN_NOSPLIT; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$O/$M" -o k --output-format csv -- python3 "$GRAFT_REPO_ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-reupload-leg > "$O/b$M.json" 2> "$O/b$M.err"
  find "$O" -name "*kernel_trace.csv" -delete
  python3 - "$O/$M" $M <<'PY'
import csv,sys,glob
f=glob.glob(sys.argv[1]+"/**/k_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "msm_window" in n or "msm_final" in n: print("%s %-30s avg=%9.1f us" % (sys.argv[2], n.split("(")[0][:30], float(r["AverageNs"])/1e3))
PY
done
