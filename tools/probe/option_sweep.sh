run() { CLIMB_AMD_OPTIONS="$1" python bench.py --no-cpu-baseline --no-cls-only-leg --steps 30 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('opt[$1]', d['ms_per_step'], d['median_ms_per_step'])"; }
for rep in 1 2; do
run ""
run "21=2"
run "21=3"
run "17=3"
run "17=4"
run "22=0"
run "22=58"
done
