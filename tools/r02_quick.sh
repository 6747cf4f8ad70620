#!/bin/bash
# quick GPU check: chosen GPU tests + the driver-shaped bench line
set -o pipefail
O="$GRAFT_REPO_ROOT/gpurun_out/${1:-r02_quick}"; mkdir -p "$O"
cd "$GRAFT_REPO_ROOT"
TESTS="${2:-tests}"
timeout -k 10 900 python -m pytest $TESTS -m gpu -x -q > "$O/gputests.log" 2>&1; rc=$?; echo "gputests rc=$rc"; tail -15 "$O/gputests.log"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > "$O/b20.json" 2> "$O/b20.err" || { tail -5 "$O/b20.err"; exit 1; }
python - "$O/b20.json" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print("steps20: value=%.3fM reupload=%.3fM ms/step=%.3f" % (d["value"]/1e6, d.get("value_reupload",0)/1e6, d["ms_per_step"]), {k: round(v,3) for k,v in d["stages_ms_one_launch_in_flight"].items()})
PY
