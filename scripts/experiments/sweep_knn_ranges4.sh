run() { python bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']; r=d['roofline']; n=[v for kk,v in k.items() if 'score norm' in kk][0]
print('$1', round(d['value']), round(d['ms_per_step'],1), 'i8 total', round(r['kernel_ms_per_step'],1), 'frac', round(r['frac'],4), 'norm', round(n['ms_per_step'],1), 'norm i8', round(n['int8_prefilter_ms_per_step'],1), 'prep', round(n['int8_preamble_ms_per_step'],1))"; }
VSC_KNN_STEP_WORK=8 VSC_KNN_STEP_MAX=524288 run work8_max512k
VSC_KNN_STEP_WORK=16 VSC_KNN_STEP_MAX=524288 run work16_max512k
VSC_KNN_STEP_WORK=16 VSC_KNN_STEP_MAX=1048576 run work16_max1M
VSC_KNN_STEP_WORK=64 VSC_KNN_STEP_MAX=1048576 run work64_max1M
VSC_KNN_STEP_WORK=64 VSC_KNN_STEP_MAX=262144 run work64_max256k
VSC_KNN_STEP_WORK=64 VSC_KNN_STEP_MAX=131072 run work64_max128k
VSC_KNN_STEP_WORK=64 VSC_KNN_STEP_MAX=65536 run work64_max64k
