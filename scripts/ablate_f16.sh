#!/bin/bash
# Timing study of sim_f16_kernel: builds libvscmi variants with parts of the K loop removed
# (VSC_F16_ABLATE bits: 1 = no barrier/DMA drain, 2 = no LDS-DMA, 4 = no fragment reads; results are wrong
# by construction) and times the sparse-candidate search with each.  Run on the GPU box:
#   bash scripts/ablate_f16.sh build   (here, cross-compiles)   /   bash scripts/ablate_f16.sh run
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  for ab in 1 2 3 4 7; do
    mkdir -p /tmp/ab$ab && cp vsc2022_amd/csrc/*.hip vsc2022_amd/csrc/*.h vsc2022_amd/csrc/Makefile /tmp/ab$ab/
    mkdir -p /tmp/include && cp include/vscmi.h /tmp/include/
    sed -i 's#../../include/vscmi.h#/tmp/include/vscmi.h#' /tmp/ab$ab/vscmi_common.h /tmp/ab$ab/Makefile
    make -C /tmp/ab$ab -j8 OUT=$PWD/build/ab/libvscmi_ab$ab.so CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DVSC_F16_ABLATE=$ab" > /dev/null
  done
else
  for ab in 0 1 2 3 4 7; do
    L=$PWD/build/ab/libvscmi_ab$ab.so; [ $ab = 0 ] && L=$PWD/vsc2022_amd/libvscmi.so
    echo -n "ablate=$ab: "; VSCMI_LIB=$L timeout -s KILL 200 python scripts/bench_sim.py --nq 65536 --nr 1000000 --K 1000 --reps 3 2>&1 | grep -o "f16 [0-9.]* ms launches=[0-9]* TFLOP/s=[0-9.]*"
  done
fi
