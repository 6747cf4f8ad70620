#!/bin/bash
# config-3 per-GPU shape (8192 proofs per step): alone and pipelined
O="$GRAFT_REPO_ROOT/gpurun_out/${1:-r02_c3}"; mkdir -p "$O"
cd "$GRAFT_REPO_ROOT"
j() { python - "$1" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print(sys.argv[1].split("/")[-1], "value=%.3fM ms/step=%.4f" % (d["value"]/1e6, d["ms_per_step"]), {k: round(v,3) for k,v in (d.get("stages_ms_one_launch_in_flight") or d["stages_ms"]).items()})
PY
}
timeout -k 10 300 python bench.py --batch 8192 --groups 1 --steps 1 --warmup 1 --no-cpu-baseline --no-reupload-leg > "$O/c3single.json" 2> "$O/c3single.err" && j "$O/c3single.json"
timeout -k 10 300 python bench.py --batch 8192 --groups 4 --steps 256 --warmup 32 --no-cpu-baseline --no-reupload-leg > "$O/c3shape.json" 2> "$O/c3shape.err" && j "$O/c3shape.json"
