#!/bin/bash
# round-2 first GPU call: the GPU suite, the driver-shaped bench with several launch shapes, then the hardware-queue probe (last:
# it is expected to abort its own process)
set -o pipefail
O="$GRAFT_REPO_ROOT/gpurun_out/r02_p1"; mkdir -p "$O"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests -m gpu -x -q > "$O/gputests.log" 2>&1; echo "gputests rc=$?" | tee -a "$O/summary.txt"
tail -5 "$O/gputests.log"
for shape in "0 0" "20 1" "10 2" "5 4" "4 5" "3 7" "2 8" "1 8"; do
  set -- $shape
  timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --groups $1 --depth $2 --no-cpu-baseline > "$O/b_g$1_d$2.json" 2> "$O/b_g$1_d$2.err" || { echo "bench $shape failed" | tee -a "$O/summary.txt"; tail -5 "$O/b_g$1_d$2.err"; exit 1; }
  python - "$O/b_g$1_d$2.json" <<'PY' | tee -a "$O/summary.txt"
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print(sys.argv[1].split("/")[-1], "value=%.3gM" % (d["value"]/1e6), "reupload=%.3gM" % (d.get("value_reupload",0)/1e6), "ms/step=%.3f" % d["ms_per_step"], "G=%d depth=%d" % (d["config"]["steps_per_launch"], d["config"]["pipeline_depth"]), {k: round(v,3) for k,v in d["stages_ms_one_launch_in_flight"].items()})
PY
done
timeout -k 10 300 python bench.py --no-cpu-baseline > "$O/b_default.json" 2> "$O/b_default.err"; python -c "
import json;d=json.loads([l for l in open('$O/b_default.json') if l.startswith('{')][0]);print('default', d['value'], d.get('value_reupload'), d['stages_ms_one_launch_in_flight'])" | tee -a "$O/summary.txt"
# hardware-queue probe: what exactly happens with GPU_MAX_HW_QUEUES=32 ("32 abort", no log kept)
cat > /tmp/hwq.py <<'PY'
import os, sys, faulthandler
faulthandler.enable()
import torch
print("queues", os.environ.get("GPU_MAX_HW_QUEUES"), flush=True)
x = torch.zeros(1024, device="cuda")
print("first alloc ok", flush=True)
streams = [torch.cuda.Stream() for _ in range(int(sys.argv[1]))]
print("streams created", len(streams), flush=True)
for i, s in enumerate(streams):
    with torch.cuda.stream(s):
        x.add_(1)
    print("launched on stream", i, flush=True)
torch.cuda.synchronize()
print("sync ok", float(x[0]), flush=True)
PY
for q in 16 24 32; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 120 python /tmp/hwq.py 40 > "$O/hwq_$q.log" 2>&1; echo "hwq $q rc=$?" | tee -a "$O/summary.txt"; tail -4 "$O/hwq_$q.log"
done
