run() { python bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']
print('$1', round(d['value']), round(d['ms_per_step'],1), 'i8', round(k['sim_i8p_kernel (int8 MFMA pre-filter, sparse batches)']['ms_per_step'],1), round(k['sim_i8p_kernel (int8 MFMA pre-filter, sparse batches)']['achieved']), 'f16', round(k['sim_f16p_kernel (fp16 MFMA pre-filter)']['ms_per_step'],1), 'resc', round(k['rescore_kernel (exact fp32 chain of the candidates)']['ms_per_step'],1), 'norm', round([v for kk,v in k.items() if 'score norm' in kk][0]['ms_per_step'],1))"; }
run base
VSC_I8P_SLICE=8 run slice8
VSC_I8P_SLICE=32 run slice32
VSC_I8P_SLICE=64 run slice64
VSC_I8_DENSITY=1e-3 run dens1e-3
VSC_I8_DENSITY=3e-3 run dens3e-3
VSC_I8_DENSITY=1e-2 run dens1e-2
