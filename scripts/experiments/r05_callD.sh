mkdir -p gpurun_out/r05d
timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_refshard.py tests/test_gpu_options.py -x -q -m gpu > gpurun_out/r05d/pytest_sharded.log 2>&1; tail -3 gpurun_out/r05d/pytest_sharded.log
for n in 2 4; do
VSC_SHARD_DEBUG=1 VSC_BENCH_SHARE_GPU=1 timeout 800 python bench.py --gpus $n --steps 2 --warmup 1 --no-extra --no-cpu-baseline > gpurun_out/r05d/share${n}_cols.json 2> gpurun_out/r05d/share${n}_cols.err
grep "^\[shard 0\]" gpurun_out/r05d/share${n}_cols.err | tail -1
python -c "
import json;d=json.loads(open('gpurun_out/r05d/share${n}_cols.json').read().strip().splitlines()[-1]);print($n, d['value'],d['ms_per_step'])"
done
