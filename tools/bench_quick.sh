python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/r2_b2.json 2>gpurun_out/r2_b2.err; python -c "
import json; d=json.load(open('gpurun_out/r2_b2.json')); print(d['value'], d['ms_per_step']); print({k:(v['ms_per_launch'],v['ms_per_step']) for k,v in d['kernels'].items() if 'pair' in k or 'trimul' in k})"
