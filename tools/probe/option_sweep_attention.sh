# attention variants (climb_set_option 12: forward query-block shape, 13: backward variant) re-checked in-step on the final tree
run() { CLIMB_AMD_OPTIONS="$1" python bench.py --no-cpu-baseline --no-cls-only-leg --steps 30 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('opt[$1]', d['ms_per_step'], d['median_ms_per_step'])"; }
for rep in 1 2; do
run ""
run "12=0"
run "12=1"
run "12=2"
run "13=0"
run "13=1"
run "13=2"
run "13=3"
run "13=4"
done
