#!/bin/bash
# in-step A/B of library options: bash tools/ab_step.sh "10=0" "10=1"   (each argument is one CLIMB_AMD_OPTIONS setting; default both TN kernels)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
[ $# -eq 0 ] && set -- "10=0" "10=1"
for opt in "$@"; do
  echo "=== CLIMB_AMD_OPTIONS=$opt"
  for i in 1 2 3; do CLIMB_AMD_OPTIONS=$opt python $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['median_ms_per_step'], d['roofline']['avg_launch_us'])"; done
done
