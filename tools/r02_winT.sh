#!/bin/bash
O="$GRAFT_REPO_ROOT/gpurun_out/r02_winT"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for T in 64 128 256; do
  export H2V_MSM_WIN_T=$T
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$O/T$T" -o k --output-format csv -- python3 "$GRAFT_REPO_ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-reupload-leg > "$O/b$T.json" 2> "$O/b$T.err"
  find "$O" -name "*kernel_trace.csv" -delete
  python3 - "$O/T$T" $T <<'PY'
import csv,sys,glob
f=glob.glob(sys.argv[1]+"/**/k_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "msm_window" in n or "msm_final" in n or "msm_fixup" in n: print("T=%s %-40s avg=%9.1f us" % (sys.argv[2], n.split("(")[0][:40], float(r["AverageNs"])/1e3))
PY
done
