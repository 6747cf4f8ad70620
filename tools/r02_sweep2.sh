#!/bin/bash
O="$GRAFT_REPO_ROOT/gpurun_out/r02_sweep2"; mkdir -p "$O"
cd "$GRAFT_REPO_ROOT"
for KB in 0 36 76; do
  H2V_FRVM_LDS_KB=$KB timeout -k 10 200 python bench.py --no-cpu-baseline --no-reupload-leg > "$O/b_$KB.json" 2> "$O/b_$KB.err" || { tail -3 "$O/b_$KB.err"; exit 1; }
  python - "$O/b_$KB.json" $KB <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print("frvm LDS KB=%s default value=%.3fM" % (sys.argv[2], d["value"]/1e6), {k: round(v,3) for k,v in d["stages_ms_one_launch_in_flight"].items()})
PY
done
