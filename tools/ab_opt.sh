#!/bin/bash
# in-step A/B of library options (GPU box): bash tools/ab_opt.sh "17=1" "17=5" ...   -- settings interleaved, two passes, bench.py's own timing
cd /root/repo
O=gpurun_out/ab_opt.txt
: > $O
for pass in 1 2; do
  for cfg in "$@"; do
    echo "== pass $pass CLIMB_AMD_OPTIONS=$cfg" >> $O
    CLIMB_AMD_OPTIONS=$cfg python bench.py --no-cpu-baseline --no-cls-only-leg --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('ms/step', j['ms_per_step'], 'median', j['median_ms_per_step'], 'NT frac', r['frac'], 'avg_launch_us', r['avg_launch_us'])
print('   ', ' | '.join(f\"{k} {v['avg_us']}\" for k,v in r['per_kind'].items()))
" >> $O
  done
done
cat $O
