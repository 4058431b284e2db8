# kernel trace + instruction counters of the end-of-round code over one bench step (run on the GPU box)
set -u
out=$PWD/gpurun_out/r05final; mkdir -p $out
export TMPDIR=/tmp
cmd="python $PWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra"
d=/tmp/kt_final; rm -rf $d
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $d -o r -- $cmd > $out/trace.log 2>&1)
db=$(find $d -name "*.db" | head -1)
[ -n "$db" ] && python scripts/rocpd_summary.py $db $out/kernel_trace.md > /dev/null
d=/tmp/pmc_final; rm -rf $d
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace -d $d -o r -- $cmd > $out/pmc_run.log 2>&1)
db=$(find $d -name "*.db" | head -1)
[ -n "$db" ] && python scripts/pmc_summary.py $db | grep "tn_pair\|select_\|^| kernel\|^|---" > $out/pmc_insts.md
head -12 $out/kernel_trace.md | cut -c1-160; cat $out/pmc_insts.md | cut -c1-200
