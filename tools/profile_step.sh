#!/bin/bash
# rocprofv3 passes behind profiles/rNN_*: bash tools/profile_step.sh r02 [precision]      (GPU box; writes gpurun_out/prof_<tag>[_<precision>]/...)
# separate passes: kernel-trace stats | PMC FETCH_SIZE | PMC WRITE_SIZE | PMC MFMA / wave-cycle counters  (never PMC together with traces other than kernel-trace)
TAG=${1:-r03}
PREC=${2:-bf16}
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/prof_$TAG
if [ "$PREC" != "bf16" ]; then O=${O}_$PREC; fi
mkdir -p $O
export CLIMB_AMD_PRECISION=$PREC          # (bench.py's --precision default)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r -- python $R/bench.py --no-cpu-baseline --no-cls-only-leg --steps 8 --warmup 3 > $O/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o r -- python $R/bench.py --no-cpu-baseline --no-cls-only-leg --steps 2 --warmup 1 > $O/pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_mfma -o r -- python $R/bench.py --no-cpu-baseline --no-cls-only-leg --steps 2 --warmup 1 > $O/pmc_mfma.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_lds -o r -- python $R/bench.py --no-cpu-baseline --no-cls-only-leg --steps 2 --warmup 1 > $O/pmc_lds.log 2>&1
python $R/bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_nocpu.json
ls -la $O
