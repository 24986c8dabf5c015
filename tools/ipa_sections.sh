#!/bin/bash
# GPU box, developer aid: average duration of the IPA training kernels with sections of k_ipa_bwd_q knocked out (GENIE_IPA_SKIP bits:
# 1 d att, 2 d logits, 4 dq, 8 dq points, 16 pair-gradient rows).  usage: [IPA_SKIPS="0"] tools/ipa_sections.sh [N] [B]
for sk in ${IPA_SKIPS:-0 1 2 4 8 16 31}; do
  GENIE_IPA_SKIP=$sk bash tools/train_profile.sh gpurun_out/ipaskip ${1:-256} ${2:-2} 0 > /dev/null 2>&1 || exit 1
  python3 - $sk <<'PY'
import csv, sys
rows = list(csv.DictReader(open('gpurun_out/ipaskip/kernel_stats.csv')))
out = {r['Name'].split('(')[0].replace('void ', ''): float(r['AverageNs']) / 1e3 for r in rows if 'k_ipa' in r['Name']}
print('skip %2s: ' % sys.argv[1] + '  '.join('%s %.1f us' % kv for kv in sorted(out.items())), flush=True)
PY
done
