#!/bin/bash
# r05: what the data-parallel path costs before the wire moves a byte -- the forced single-rank RCCL step (CLIMB_AMD_FORCE_DDP=1), collectives under the
# backward and deferred, against the plain step on the same box; then kernel traces of the plain and the faster forced step, differenced
# (tools/trace_diff.py).  bash tools/ddp_trace.sh   (GPU box; writes gpurun_out/forced_ddp_trace_diff.txt)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out
ms() { tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
{
echo "forced single-rank RCCL step (CLIMB_AMD_FORCE_DDP=1) against the plain step, same box; unprofiled ms/step, bench.py --steps 20, two passes:"
for pass in 1 2; do
  echo "  plain                                  $(python $R/bench.py --no-cpu-baseline --no-cls-only-leg 2>/dev/null | ms)"
  echo "  forced, collectives under the backward $(CLIMB_AMD_FORCE_DDP=1 CLIMB_AMD_DP_OVERLAP=1 python $R/bench.py --no-cpu-baseline --no-cls-only-leg 2>/dev/null | ms)"
  echo "  forced, collectives after the backward $(CLIMB_AMD_FORCE_DDP=1 CLIMB_AMD_DP_OVERLAP=0 python $R/bench.py --no-cpu-baseline --no-cls-only-leg 2>/dev/null | ms)"
done
} > $O/forced_ddp_trace_diff.txt
rm -rf $O/prof_plain $O/prof_ddp1 $O/prof_ddp0
rocprofv3 --kernel-trace --output-format csv -d $O/prof_plain -o r -- python $R/bench.py --no-cpu-baseline --no-cls-only-leg --steps 8 --warmup 3 > $O/prof_plain.log 2>&1
CLIMB_AMD_FORCE_DDP=1 CLIMB_AMD_DP_OVERLAP=1 rocprofv3 --kernel-trace --output-format csv -d $O/prof_ddp1 -o r -- python $R/bench.py --no-cpu-baseline --no-cls-only-leg --steps 8 --warmup 3 > $O/prof_ddp1.log 2>&1
CLIMB_AMD_FORCE_DDP=1 CLIMB_AMD_DP_OVERLAP=0 rocprofv3 --kernel-trace --output-format csv -d $O/prof_ddp0 -o r -- python $R/bench.py --no-cpu-baseline --no-cls-only-leg --steps 8 --warmup 3 > $O/prof_ddp0.log 2>&1
cd $O
{
echo; echo "== collectives AFTER the backward (one grouped weight-gradient launch, one collective) against the plain step"
python $R/tools/trace_diff.py prof_ddp0/r_kernel_trace.csv 5 prof_plain/r_kernel_trace.csv 5
echo; echo "== collectives UNDER the backward (groups of 4 layers) against the plain step"
python $R/tools/trace_diff.py prof_ddp1/r_kernel_trace.csv 5 prof_plain/r_kernel_trace.csv 5
} >> $O/forced_ddp_trace_diff.txt
cat $O/forced_ddp_trace_diff.txt
