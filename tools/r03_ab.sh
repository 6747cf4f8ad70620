# A/B of forced kernel variants at the driver's command (bench.py --tuning reaches the headline context since this script exists)
mkdir -p gpurun_out/r03_ab
for t in "none" "msm_no_term_split=1" "msm_global_sort=1" "msm_window_threads=256" "msm_window_wpw=2" "msm_acc_waves=4" "none2"; do
  f=gpurun_out/r03_ab/$(echo $t | tr '=,' '__').json
  case $t in none*) T="";; *) T="--tuning $t";; esac
  timeout -k 10 120 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-reupload-leg --no-extra-legs $T > $f 2> ${f%.json}.err || exit 1
done
for m in 1 2 3; do
  f=gpurun_out/r03_ab/upload_mode_$m.json
  timeout -k 10 120 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --tuning upload_mode=$m > $f 2> ${f%.json}.err || exit 1
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_ab/*.json')):
    for l in open(f):
        if l.startswith('{'):
            j=json.loads(l); print(f.split('/')[-1], j.get('tuning'), round(j['value']/1e6,3), round(j.get('value_reupload',0)/1e6,3), {k:round(v,3) for k,v in j['stages_ms'].items()})
PY
