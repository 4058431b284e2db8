#!/bin/bash
# A/B of the paired int8 shape's knobs on configs[3] (same box): slice size, ring depth (a second library), pairs off
run() { python bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']; r=d['roofline']
print('$1', round(d['value']), round(d['ms_per_step'],1), 'i8 total', round(r['kernel_ms_per_step'],1), 'frac', round(r['frac'],4), 'cand', r['prefilter_candidates_last_search'], 'resc', round(k['rescore_kernel (exact fp32 chain of the candidates)']['ms_per_step'],1), 'norm', round([v for kk,v in k.items() if 'score norm' in kk][0]['ms_per_step'],1))"; }
run base
VSC_I8P_PAIR=0 run pair0
VSC_I8P_SLICE=16 run slice16
VSC_I8P_SLICE=64 run slice64
VSCMI_LIB=$PWD/build/libvscmi_pf2_4.so run ring4
VSCMI_LIB=$PWD/build/libvscmi_pf2_4.so VSC_I8P_SLICE=16 run ring4_slice16
run base_again
