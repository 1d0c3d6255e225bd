#!/bin/bash
# per-kernel time of one step in a given precision mode: bash tools/r06_fp32_trace.sh <precision> [steps]   (GPU box)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
P=${1:-fp32}
O=$R/gpurun_out/prof_mode_$P
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o r -- python $R/bench.py --precision $P --no-cpu-baseline --no-cls-only-leg --steps ${2:-3} --warmup 2 > $O/log 2>&1
tail -3 $O/log
python - <<PY
import csv, re, collections
rows=list(csv.DictReader(open("$O/r_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:40]:
    n=re.sub(r"\(.*","",r["Name"]).replace("void ","")[:70]
    print(f'{float(r["TotalDurationNs"])/1e6:9.3f} ms {int(r["Calls"]):6d} calls avg {float(r["AverageNs"])/1e3:9.1f} us  {100*float(r["TotalDurationNs"])/tot:5.1f}%  {n}')
print("total kernel ms", tot/1e6)
PY
