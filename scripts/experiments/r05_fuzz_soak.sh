# the fuzz soaks at the round's HEAD ("Next round" 2c); results -> gpurun_out/r05fuzz/soak.log
mkdir -p gpurun_out/r05fuzz; L=gpurun_out/r05fuzz/${1:-soak}.log; : > $L
run() { echo "\$ $*" >> $L; ( timeout 600 env "$@" 2>&1 | grep -v amdgpu.ids | tail -2 ) >> $L; }
run VSC_I8=2 python scripts/fuzz_prefilter.py --seconds 200 --seed 801
run VSC_I8=2 python scripts/fuzz_prefilter.py --seconds 120 --seed 802 --big
run python scripts/fuzz_prefilter.py --seconds 100 --seed 803
run VSC_I8=2 VSC_I8P_PAIR=2 python scripts/fuzz_prefilter.py --seconds 100 --seed 804
run python scripts/fuzz_pipeline.py --seconds 100 --seed 805
run python scripts/fuzz_sharded.py --seconds 200 --seed 806
run VSC_SHARD_MODE=rows python scripts/fuzz_sharded.py --seconds 100 --seed 807
cat $L
