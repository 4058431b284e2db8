run() { python bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']; r=d['roofline']; n=[v for kk,v in k.items() if 'score norm' in kk][0]
print('$1', round(d['value']), round(d['ms_per_step'],1), 'i8 total', round(r['kernel_ms_per_step'],1), 'norm', round(n['ms_per_step'],1), 'norm i8', round(n['int8_prefilter_ms_per_step'],1), 'prep', round(n['int8_preamble_ms_per_step'],1))"; }
run base
VSC_KNN_S0MIN=1024 run s0min1024
VSC_KNN_S0MIN=1024 VSC_KNN_RATIO=3 run s0min1024_ratio3
VSC_KNN_S0MIN=512 VSC_KNN_RATIO=4 run s0min512_ratio4
VSC_KNN_S0MIN=512 VSC_KNN_RATIO=3 run s0min512_ratio3
VSC_KNN_S0MIN=256 VSC_KNN_RATIO=4 run s0min256_ratio4
VSC_KNN_S0MIN=2048 VSC_KNN_RATIO=3 run s0min2048_ratio3
