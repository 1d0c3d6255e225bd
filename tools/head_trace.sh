#!/bin/bash
# per-launch timeline of the pooler / task-head chain inside the step (between the last encoder layer's forward and the first backward GEMM):
#   bash tools/head_trace.sh            (GPU box; CLIMB_AMD_SKINNY_HEADS=0 for the r01-r03 launches)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/prof_head_${CLIMB_AMD_SKINNY_HEADS:-1}
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -o r -- python $R/bench.py --no-cpu-baseline --no-cls-only-leg --steps 6 --warmup 3 > $O/log 2>&1
python - <<PY
import csv, re
rows=list(csv.DictReader(open("$O/r_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=[re.sub(r"\(.*","",r["Kernel_Name"]).replace("void ","") for r in rows]
idx=[i for i,n in enumerate(names) if "tn_grouped" in n]
a,b=idx[4],idx[5]
seq=list(zip(names[a+1:b+1], rows[a+1:b+1]))
# the chain: from the launch after the last attn_fwd's two GEMM successors to the first DGELU-epilogue GEMM
last_fwd=max(i for i,(n,_) in enumerate(seq) if "attn_fwd" in n)
first_bwd=min(i for i,(n,_) in enumerate(seq) if i>last_fwd and "attn_bwd" in n)
lo=last_fwd+5; hi=first_bwd-4
t0=int(seq[lo][1]["Start_Timestamp"]); tot=0
for n,r in seq[lo:hi]:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"]); tot+=(e-s)/1e3
    print(f"{(s-t0)/1e3:8.1f} {(e-s)/1e3:7.1f}  {n[:70]}")
print(f"{hi-lo} launches, kernel time {tot:.1f} us, span {(int(seq[hi-1][1]['End_Timestamp'])-t0)/1e3:.1f} us")
PY
