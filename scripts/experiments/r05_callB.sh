mkdir -p gpurun_out/r05b
cd scripts/ubench
for dd in "0 0" "1 1" "2 2" "3 3" "4 0" "5 5" "0 0"; do set -- $dd; I8_DATA_Q=$1 I8_DATA_R=$2 timeout 120 ./i8_tiles 32768 1000000 6 >> ../../gpurun_out/r05b/i8_data_power.log 2>&1; done
cd ../..
timeout 600 python -m pytest tests/test_gpu_topk_proven.py tests/test_gpu_sharded.py tests/test_gpu_options.py tests/test_gpu_refshard.py tests/test_gpu_fullsize.py tests/test_gpu_configs_fullsize.py -q --durations=25 -m gpu > gpurun_out/r05b/durations.log 2>&1
VSC_BENCH_SHARE_GPU=1 timeout 500 python bench.py --gpus 2 --steps 3 --warmup 1 --no-extra --no-cpu-baseline > gpurun_out/r05b/share2.json 2> gpurun_out/r05b/share2.err
bash scripts/pmc_bench.sh r05pmc > gpurun_out/r05b/pmc_bench.log 2>&1
tail -3 gpurun_out/r05b/i8_data_power.log; tail -c 600 gpurun_out/r05b/share2.json
