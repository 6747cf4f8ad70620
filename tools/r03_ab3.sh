# window shapes at the driver's command after the barrier-free tree
mkdir -p gpurun_out/r03_ab3
for t in "none" "msm_window_threads=64,msm_window_wpw=1" "msm_window_threads=64,msm_window_wpw=4" "msm_window_threads=128,msm_window_wpw=1" "msm_window_threads=256,msm_window_wpw=1" "msm_window_threads=64,msm_window_wpw=2" "none2"; do
  f=gpurun_out/r03_ab3/$(echo $t | tr '=,' '__').json
  case $t in none*) T="";; *) T="--tuning $t";; esac
  timeout -k 10 120 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-reupload-leg --no-extra-legs $T > $f 2> ${f%.json}.err || exit 1
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_ab3/*.json')):
    for l in open(f):
        if l.startswith('{'):
            j=json.loads(l); print(f.split('/')[-1], j.get('tuning'), round(j['value']/1e6,3), {k:round(v,3) for k,v in j['stages_ms'].items()})
PY
