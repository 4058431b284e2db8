#!/bin/bash
# PMC passes over a micro-benchmark binary (run on the GPU box): one rocprofv3 run per counter group
# (never combined with the hip/hsa trace domains), summaries under gpurun_out/<tag>/.
#   bash scripts/pmc_ubench.sh <tag> <binary> [args...]
set -u
tag=$1; shift
out=$PWD/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU" \
           "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  d=/tmp/pmc_${tag}_$i; rm -rf $d
  bin=$(readlink -f "$1"); shift; set -- "$bin" "$@"; (cd /tmp && timeout 600 rocprofv3 --pmc $grp --kernel-trace -d $d -o r -- "$@" > $out/run_$i.log 2>&1)
  db=$(find $d -name "*.db" | head -1)
  echo "## $grp" >> $out/pmc.md
  if [ -n "$db" ]; then python $PWD/scripts/pmc_summary.py $db >> $out/pmc.md; else echo "(no db; see run_$i.log)" >> $out/pmc.md; tail -5 $out/run_$i.log >> $out/pmc.md; fi
  echo >> $out/pmc.md
done
cat $out/pmc.md
