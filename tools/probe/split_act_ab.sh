#!/bin/bash
# GELU / x GELU' as epilogues of the split NT launch (CLIMB_AMD_SPLIT_FUSED_ACT=1) against the separate passes: step time and the two launches
cd /root/repo
for f in 0 1; do
  echo "== CLIMB_AMD_SPLIT_FUSED_ACT=$f"
  CLIMB_AMD_SPLIT_FUSED_ACT=$f python bench.py --precision bf16x3 --no-cpu-baseline --no-cls-only-leg --steps 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms/step', d['ms_per_step'], 'samples/s', d['value'])
for k, v in d['roofline']['per_kind'].items():
    if 'N3072' in k or 'epi1' in k: print('  ', k, v['avg_us'], 'us')
"
done
