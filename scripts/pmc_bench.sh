#!/bin/bash
# PMC passes + kernel trace over one bench.py step (run on the GPU box).  One rocprofv3 run per counter group, never
# combined with the hip/hsa trace domains.  Summaries land in gpurun_out/<tag>/ ; copy what is cited into profiles/.
#   bash scripts/pmc_bench.sh <tag> [bench.py args...]
set -u
tag=$1; shift
out=$PWD/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
cmd="python $PWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra $*"
d=/tmp/kt_$tag; rm -rf $d
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $d -o r -- $cmd > $out/trace.log 2>&1)
db=$(find $d -name "*.db" | head -1)
[ -n "$db" ] && python scripts/rocpd_summary.py $db $out/kernel_trace.md > /dev/null && python scripts/rocpd_dispatches.py $db sim_f16 > $out/dispatches_sim_f16.md && python scripts/rocpd_dispatches.py $db sim_i8p > $out/dispatches_sim_i8p.md
grep '^{"metric"' $out/trace.log > $out/bench_line_under_rocprof.json
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE" \
           "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  d=/tmp/pmc_${tag}_$i; rm -rf $d
  (cd /tmp && timeout 900 rocprofv3 --pmc $grp --kernel-trace -d $d -o r -- $cmd > $out/pmc_run_$i.log 2>&1)
  db=$(find $d -name "*.db" | head -1)
  echo "## $grp" >> $out/pmc.md
  if [ -n "$db" ]; then python scripts/pmc_summary.py $db | grep -v "select_\|pack_\|_hits_kernel\|pair_.*_kernel\|rocprim\|tail_reset\|knn_\|row_norm" >> $out/pmc.md; else echo "(no db)" >> $out/pmc.md; tail -3 $out/pmc_run_$i.log >> $out/pmc.md; fi
  echo >> $out/pmc.md
done
