#!/bin/bash
# per-kernel time of the adapter step (configs[2]) against the plain step: bash tools/r06_adapter_trace.sh   (GPU box)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for mode in plain adapter; do
  O=$R/gpurun_out/prof_adapter_$mode
  rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o r -- python $R/tools/probe/adapter_step.py $mode 8 > $O/log 2>&1
  tail -1 $O/log
  python - <<PY
import csv, re
rows=list(csv.DictReader(open("$O/r_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("== $mode: kernel ms per step (11 steps)", round(tot/1e6/11,3))
for r in rows[:26]:
    n=re.sub(r"\(.*","",r["Name"]).replace("void ","")[:80]
    print(f'{float(r["TotalDurationNs"])/1e6/11:8.3f} ms/step {int(r["Calls"])/11:7.1f} calls/step avg {float(r["AverageNs"])/1e3:8.1f} us  {n}')
PY
done
