#!/bin/bash
O="$GRAFT_REPO_ROOT/gpurun_out/${1:-r02_sweep}"; mkdir -p "$O"
cd "$GRAFT_REPO_ROOT"
for T in 64 128 256; do
  H2V_MSM_WIN_T=$T timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-reupload-leg > "$O/b_T$T.json" 2> "$O/b_T$T.err" || { tail -3 "$O/b_T$T.err"; exit 1; }
  python - "$O/b_T$T.json" $T <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print("T=%s value=%.3fM" % (sys.argv[2], d["value"]/1e6), {k: round(v,3) for k,v in d["stages_ms_one_launch_in_flight"].items()})
PY
done
