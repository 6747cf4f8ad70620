#!/bin/bash
# split-Horner parts sweep on the driver-shaped launch
O="$GRAFT_REPO_ROOT/gpurun_out/${1:-r02_parts}"; mkdir -p "$O"
cd "$GRAFT_REPO_ROOT"
for P in 4 6 8; do
  H2V_MSM_PARTS=$P timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-reupload-leg > "$O/b20_p$P.json" 2> "$O/b20_p$P.err" || { tail -5 "$O/b20_p$P.err"; exit 1; }
  python - "$O/b20_p$P.json" $P <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print("parts", sys.argv[2], "steps20: value=%.3fM" % (d["value"]/1e6), {k: round(v,3) for k,v in d["stages_ms_one_launch_in_flight"].items()})
PY
done
H2V_MSM_PARTS=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-reupload-leg > "$O/bdef_p1.json" 2> "$O/bdef_p1.err" && grep -o '"value": [0-9.]*' "$O/bdef_p1.json" | head -1
H2V_MSM_PARTS=4 timeout -k 10 300 python bench.py --no-cpu-baseline --no-reupload-leg > "$O/bdef_p4.json" 2> "$O/bdef_p4.err" && grep -o '"value": [0-9.]*' "$O/bdef_p4.json" | head -1
