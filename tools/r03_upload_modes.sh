mkdir -p gpurun_out/r03_j
for m in 1 2 3; do timeout -k 10 150 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --tuning upload_mode=$m > gpurun_out/r03_j/b20_m$m.json 2> gpurun_out/r03_j/b20_m$m.err || exit 1; done
timeout -k 10 150 python bench.py --no-cpu-baseline --no-extra-legs --tuning upload_mode=2 > gpurun_out/r03_j/bdef_m2.json 2> gpurun_out/r03_j/bdef_m2.err
