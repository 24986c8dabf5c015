#!/bin/bash
# GPU box: bench each ablation library, print per-kernel ms.  Usage: tools/abl_run.sh 1 3 4
cd "$GRAFT_REPO_ROOT"
export GENIE_MATH=hx
for n in 0 "$@"; do
  if [ "$n" = 0 ]; then unset GENIE_HIP_LIB; else export GENIE_HIP_LIB=$PWD/genie2_amd/lib/abl/libgenie_abl$n.so; fi
  timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/abl$n.json 2> gpurun_out/abl$n.err || echo "abl $n failed"
  python - <<PY
import json
d = json.load(open('gpurun_out/abl$n.json'))
print('abl $n', round(d['value'], 2), {k: round(v['ms_per_step'], 3) for k, v in d['kernels'].items() if v['ms_per_step'] > 0.5})
PY
done
