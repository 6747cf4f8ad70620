#!/bin/bash
# driver-shaped run (20 steps) cut into launches in different ways
O="$GRAFT_REPO_ROOT/gpurun_out/${1:-r02_shape}"; mkdir -p "$O"
cd "$GRAFT_REPO_ROOT"
for S in "0 0" "10 2" "5 4" "4 5" "7 3"; do
  set -- $S
  timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --groups $1 --depth $2 --no-cpu-baseline --no-reupload-leg > "$O/b_$1_$2.json" 2> "$O/b_$1_$2.err" || { tail -5 "$O/b_$1_$2.err"; exit 1; }
  python - "$O/b_$1_$2.json" $1 $2 <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print("groups", sys.argv[2], "depth", sys.argv[3], "value=%.3fM ms/step=%.4f" % (d["value"]/1e6, d["ms_per_step"]))
PY
done
