#!/bin/bash
# ratio / first-subset sweep of the 1-NN's reference ranges after the launches over short ranges grew (configs[3] step)
run() { python bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']; r=d['roofline']; n=[v for kk,v in k.items() if 'score norm' in kk][0]
print('$1', round(d['value']), round(d['ms_per_step'],1), 'i8 total', round(r['kernel_ms_per_step'],1), 'norm', round(n['ms_per_step'],1), 'norm i8', round(n['int8_prefilter_ms_per_step'],1), 'prep', round(n['int8_preamble_ms_per_step'],1))"; }
run base
VSC_KNN_RATIO=3 run ratio3
VSC_KNN_RATIO=2 run ratio2
VSC_KNN_RATIO=2.5 run ratio2.5
VSC_KNN_S0MIN=2048 run s0min2048
VSC_KNN_S0MIN=2048 VSC_KNN_RATIO=3 run s0min2048_ratio3
VSC_KNN_S0MIN=8192 run s0min8192
run base_again
