#!/bin/bash
# the measurements quoted in DESIGN.md / README.md for round 2 (one GPU)
O="$GRAFT_REPO_ROOT/gpurun_out/${1:-r02_numbers}"; mkdir -p "$O"
cd "$GRAFT_REPO_ROOT"
j() { python - "$1" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print(sys.argv[1].split("/")[-1], "value=%.3fM reupload=%.3fM ms/step=%.4f n_gpus=%d" % (d["value"]/1e6, d.get("value_reupload",0)/1e6, d["ms_per_step"], d["n_gpus"]), {k: round(v,3) for k,v in (d.get("stages_ms_one_launch_in_flight") or d["stages_ms"]).items()}, d.get("config3"), (d.get("cpu_baseline") or {}).get("value"))
PY
}
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > "$O/driver.json" 2> "$O/driver.err" && j "$O/driver.json"
timeout -k 10 300 python bench.py --no-cpu-baseline > "$O/default.json" 2> "$O/default.err" && j "$O/default.json"
timeout -k 10 300 python bench.py --steps 1 --warmup 1 --groups 1 --no-cpu-baseline --no-reupload-leg > "$O/single.json" 2> "$O/single.err" && j "$O/single.json"
timeout -k 10 300 python bench.py --batch 8192 --groups 4 --steps 256 --warmup 32 --no-cpu-baseline --no-reupload-leg > "$O/c3shape.json" 2> "$O/c3shape.err" && j "$O/c3shape.json"
timeout -k 10 300 python bench.py --batch 8192 --groups 1 --steps 1 --warmup 1 --no-cpu-baseline --no-reupload-leg > "$O/c3single.json" 2> "$O/c3single.err" && j "$O/c3single.json"
H2V_BENCH_BACKEND=gloo H2V_BENCH_ONE_DEVICE=1 timeout -k 10 600 python bench.py --gpus 2 --steps 20 --warmup 5 --config3-steps 4 > "$O/two_ranks_one_gpu.json" 2> "$O/two_ranks_one_gpu.err" && j "$O/two_ranks_one_gpu.json"
for v in "shplonk blake2b" "gwc blake2b" "shplonk keccak256" "gwc keccak256"; do set -- $v; timeout -k 10 300 python tools/bench_wide.py --circuit vector_mul --k 8 --groups 32 --steps 32 --multiopen $1 --transcript $2 2>&1 | tail -1 | cut -c1-200 | tee -a "$O/instantiations.txt"; done
timeout -k 10 600 python tools/bench_wide.py --circuit wide --k 10 --groups 8 --steps 32 2>&1 | tail -1 | cut -c1-400 | tee "$O/wide.txt"
timeout -k 10 600 python tools/bench_wide.py --circuit wide --k 10 --groups 1 --depth 1 --steps 4 2>&1 | tail -1 | cut -c1-400 | tee "$O/wide_single.txt"
