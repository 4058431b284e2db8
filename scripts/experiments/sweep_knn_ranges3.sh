run() { python bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']; r=d['roofline']; n=[v for kk,v in k.items() if 'score norm' in kk][0]
print('$1', round(d['value']), round(d['ms_per_step'],1), 'i8 total', round(r['kernel_ms_per_step'],1), 'norm', round(n['ms_per_step'],1), 'norm i8', round(n['int8_prefilter_ms_per_step'],1), 'prep', round(n['int8_preamble_ms_per_step'],1))"; }
VSC_KNN_S0MIN=4096 run s0_4096_work1
VSC_KNN_S0MIN=4096 VSC_KNN_STEP_WORK=2 run s0_4096_work2
VSC_KNN_S0MIN=4096 VSC_KNN_STEP_WORK=4 run s0_4096_work4
VSC_KNN_S0MIN=4096 VSC_KNN_STEP_WORK=8 VSC_KNN_STEP_MAX=524288 run s0_4096_work8
run s0_3584_work1
VSC_KNN_STEP_WORK=2 run s0_3584_work2
VSC_KNN_STEP_WORK=4 run s0_3584_work4
