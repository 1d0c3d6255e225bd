#!/bin/bash
# r05 A/B (GPU box): what the up-projection saves for the backward (CLIMB_AMD_GELU_SAVE=pre|deriv) x which kernel runs the GELU kinds
# (climb_set_option 17: 1 = 8-wave kernel for GELU / GELUD, 3 = the four-wave kernel for every epilogue); settings interleaved, two passes.
cd /root/repo
O=gpurun_out/ab_gelu.txt
: > $O
for pass in 1 2; do
  for cfg in "pre 17=1" "deriv 17=1" "pre 17=3" "deriv 17=3"; do
    set -- $cfg
    echo "== pass $pass GELU_SAVE=$1 OPTIONS=$2" >> $O
    CLIMB_AMD_GELU_SAVE=$1 CLIMB_AMD_OPTIONS=$2 python bench.py --no-cpu-baseline --no-cls-only-leg --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('ms/step', j['ms_per_step'], 'median', j['median_ms_per_step'], 'NT frac', r['frac'], 'avg_launch_us', r['avg_launch_us'])
for k,v in r['per_kind'].items():
    if 'epi1' in k or 'epi3' in k or 'epi8' in k or 'epi9' in k: print('   ', k, v['avg_us'])
" >> $O
  done
done
cat $O
