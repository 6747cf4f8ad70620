#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run() { echo "== $*"; env "$@" timeout -k 10 500 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 1; }
run H2V_FRVM_STREAMS=2
run H2V_FRVM_STREAMS=3
run H2V_MSM_PARTS=4
run H2V_PAIRING_ONE_STREAM=1 H2V_FRVM_ONE_STREAM=1
run H2V_MSM_NO_TERM_SPLIT=1 H2V_MSM_GLOBAL_SORT=1
