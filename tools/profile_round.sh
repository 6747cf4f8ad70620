#!/bin/bash
# Collect the rocprofv3 evidence for a round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <out_dir_under_gpurun_out>
# Passes (never --pmc together with a trace domain; the program itself follows `--`):
#   1. --kernel-trace --stats of the DRIVER's command:  python3 bench.py --gpus 1 --steps 20 --warmup 5
#   2. --kernel-trace --stats of the default command:   python3 bench.py              (32 steps per launch, 8 launches in flight)
#   3. --pmc FETCH_SIZE                                  (driver command, no CPU baseline / re-upload leg: same GPU launches)
#   4. --pmc WRITE_SIZE
#   5. --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY
#   6-8. the three counter passes for one 32-step launch in flight (--steps 128 --depth 1): the shape of the throughput run
# tools/summarize_profiles.py turns the raw CSVs into the files committed under profiles/.
set -e
OUT="$GRAFT_REPO_ROOT/gpurun_out/${1:-prof}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py"
DRV="--gpus 1 --steps 20 --warmup 5"
LEAN="--no-cpu-baseline --no-reupload-leg --no-extra-legs"
D1="--steps 128 --warmup 32 --depth 1 $LEAN"
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d "$OUT/driver" -o k --output-format csv -- python3 "$B" $DRV > "$OUT/driver.json" 2> "$OUT/driver.err"
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d "$OUT/default" -o k --output-format csv -- python3 "$B" --no-cpu-baseline --no-extra-legs > "$OUT/default.json" 2> "$OUT/default.err"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d "$OUT/drv_fetch" -o k --output-format csv -- python3 "$B" $DRV $LEAN > "$OUT/drv_fetch.json" 2> "$OUT/drv_fetch.err"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE -d "$OUT/drv_write" -o k --output-format csv -- python3 "$B" $DRV $LEAN > "$OUT/drv_write.json" 2> "$OUT/drv_write.err"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY -d "$OUT/drv_valu" -o k --output-format csv -- python3 "$B" $DRV $LEAN > "$OUT/drv_valu.json" 2> "$OUT/drv_valu.err"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d "$OUT/d1_fetch" -o k --output-format csv -- python3 "$B" $D1 > "$OUT/d1_fetch.json" 2> "$OUT/d1_fetch.err"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE -d "$OUT/d1_write" -o k --output-format csv -- python3 "$B" $D1 > "$OUT/d1_write.json" 2> "$OUT/d1_write.err"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY -d "$OUT/d1_valu" -o k --output-format csv -- python3 "$B" $D1 > "$OUT/d1_valu.json" 2> "$OUT/d1_valu.err"
# the raw per-dispatch traces are large: keep the stats and the counter tables — and the driver command's trace (a few thousand rows),
# from which summarize_profiles.py takes the headline launches alone (the command also runs legs on other VKs under the same kernel names)
find "$OUT" -name "*kernel_trace.csv" -not -path "*/driver/*" -delete
ls -R "$OUT" | head -60
