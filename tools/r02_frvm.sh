#!/bin/bash
# two-stream Fr program: schedule statistics and kernel time for a few values of the scheduler's barrier charge
cd "$GRAFT_REPO_ROOT"
for S in 0.3 1 2 4 8; do
  echo "bar $S"
  H2V_FRVM_SLACK=1 H2V_FRVM_BAR=$S H2V_DUMP_PLAN=1 python bench.py --steps 1 --warmup 0 --groups 1 --no-cpu-baseline --no-reupload-leg 2>&1 | grep "h2v plan. stream" | tail -2
  H2V_FRVM_SLACK=1 H2V_FRVM_BAR=$S bash tools/r02_kstats.sh r02_kf$S 20 | grep -E "frvm"
done
