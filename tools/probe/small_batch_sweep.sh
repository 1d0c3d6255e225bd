# per-GPU batches of the strong-scaling launch (global batch 64 over 8 / 4 / 2 GPUs): which NT kernel family serves M = 1536 / 3072 / 6144 rows best?
run() { CLIMB_AMD_OPTIONS="$2" python bench.py --batch $1 --no-cpu-baseline --no-cls-only-leg --steps 30 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $1 opt[$2]', d['ms_per_step'], d['median_ms_per_step'])"; }
for b in 8 16 32; do
run $b ""
run $b "17=0"
run $b "17=0,7=0"
run $b "17=0,7=0,5=0"
run $b "17=0,7=0,5=0,2=1"
done
