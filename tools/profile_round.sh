#!/bin/bash
# Collect the rocprofv3 evidence for a round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <out_dir_under_gpurun_out>
# Passes (never --pmc together with a trace domain):
#   1. --kernel-trace --stats of the default bench command
#   2. --kernel-trace --stats of one launch in flight (--depth 1): undisturbed per-kernel durations
#   3. --pmc FETCH_SIZE           (depth 1)        4. --pmc WRITE_SIZE (depth 1)
#   5. --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY (depth 1)
# tools/summarize_profiles.py turns the raw CSVs into the files committed under profiles/.
set -e
OUT="$GRAFT_REPO_ROOT/gpurun_out/${1:-prof}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
D1="--steps 128 --warmup 32 --depth 1 --no-cpu-baseline"
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d "$OUT/default" -o k --output-format csv -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline > "$OUT/default.json" 2> "$OUT/default.err"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$OUT/depth1" -o k --output-format csv -- python3 "$GRAFT_REPO_ROOT/bench.py" $D1 > "$OUT/depth1.json" 2> "$OUT/depth1.err"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o k --output-format csv -- python3 "$GRAFT_REPO_ROOT/bench.py" $D1 > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" -o k --output-format csv -- python3 "$GRAFT_REPO_ROOT/bench.py" $D1 > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY -d "$OUT/pmc_valu" -o k --output-format csv -- python3 "$GRAFT_REPO_ROOT/bench.py" $D1 > "$OUT/pmc_valu.json" 2> "$OUT/pmc_valu.err"
# the raw per-dispatch traces are large: keep the stats and the counter tables only
find "$OUT" -name "*kernel_trace.csv" -delete
ls -R "$OUT" | head -40
