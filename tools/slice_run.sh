#!/bin/bash
# GPU box: bench the hx path for several batch-slice sizes of the triangle multiplication.
cd "$GRAFT_REPO_ROOT"
for sb in "$@"; do
  GENIE_HX_SLICE=$sb timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/slice$sb.json 2> gpurun_out/slice$sb.err || echo "slice $sb failed"
  python - <<PY
import json
d = json.load(open('gpurun_out/slice$sb.json'))
print('slice $sb', round(d['value'], 2), {k: round(v['ms_per_step'], 3) for k, v in d['kernels'].items() if v['ms_per_step'] > 0.5})
PY
done
