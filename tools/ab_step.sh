cd /tmp && export TMPDIR=/tmp
R=/root/repo
mkdir -p $R/gpurun_out/r2g
for opt in "7=0" "7=1"; do
  echo "=== CLIMB_AMD_OPTIONS=$opt"
  for i in 1 2; do CLIMB_AMD_OPTIONS=$opt python $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['avg_launch_us'])"; done
  CLIMB_AMD_OPTIONS=$opt rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2g/prof_$opt -o r -- python $R/bench.py --no-cpu-baseline --steps 8 --warmup 3 > /dev/null 2>&1
  f=$(find $R/gpurun_out/r2g/prof_$opt -name "*kernel_stats.csv" | head -1)
  head -25 $f | cut -c1-150
done
