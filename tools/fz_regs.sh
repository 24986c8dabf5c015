#!/bin/bash
# Developer aid: VGPRs / spills / scratch of the fused chains for a set of -D flags.   tools/fz_regs.sh [-DFZ_ASYM=1 ...]
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGENIE_BUILD "$@" -S --cuda-device-only -o /tmp/fz_regs.s genie2_amd/csrc/pair_fused_kernels.hip 2>/dev/null
python3 - <<'PY'
import re
txt = open('/tmp/fz_regs.s').read()
for m in re.finditer(r'\.name:\s+(_Z\S*k_pair_fused\S*)\n(.*?)\.wavefront_size', txt, re.S):
    body = m.group(2)
    g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, body)
    print(m.group(1)[:60], 'vgpr', g('vgpr_count').group(1), 'spill', g('vgpr_spill_count').group(1), 'sgpr', g('sgpr_count').group(1),
          'scratch', g('private_segment_fixed_size').group(1))
PY
