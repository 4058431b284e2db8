#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes over one bench.py step (run on the GPU box): the `roofline.traffic` figure.
# One rocprofv3 run per counter (they do not fit one pass; never combined with the hip/hsa trace domains).
#   bash scripts/pmc_traffic.sh <tag> [bench.py args...]
set -u
tag=$1; shift
out=$PWD/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
cmd="python $PWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra $*"
for grp in FETCH_SIZE WRITE_SIZE; do
  for attempt in 1 2 3; do
    d=/tmp/pmc_${tag}_$grp; rm -rf $d
    (cd /tmp && timeout 900 rocprofv3 --pmc $grp --kernel-trace -d $d -o r -- $cmd > $out/pmc_run_$grp.log 2>&1)
    db=$(find $d -name "*.db" | head -1)
    [ -n "$db" ] && break
  done
  echo "## $grp" >> $out/traffic.md
  if [ -n "$db" ]; then python scripts/pmc_summary.py $db | grep "sim_i8p\|sim_f16p\|rescore_dense\|^| kernel\|^|---" >> $out/traffic.md; else echo "(no db)" >> $out/traffic.md; tail -3 $out/pmc_run_$grp.log >> $out/traffic.md; fi
  echo >> $out/traffic.md
done
cat $out/traffic.md
