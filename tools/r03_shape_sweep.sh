mkdir -p gpurun_out/r03_g
for gd in "20 1" "10 2" "7 3" "5 4" "4 5"; do set -- $gd; timeout -k 10 120 python bench.py --gpus 1 --steps 20 --warmup 5 --groups $1 --depth $2 --no-cpu-baseline --no-reupload-leg --no-extra-legs > gpurun_out/r03_g/g$1_d$2.json 2> gpurun_out/r03_g/g$1_d$2.err || exit 1; done
