#!/bin/bash
# per-kernel times of the NT GEMMs INSIDE the step, by position in the layer: bash tools/step_trace.sh [CLIMB_AMD_OPTIONS]   (GPU box)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/prof_trace_$(echo "${1:-default}" | tr '=,' '__')
rm -rf $O; mkdir -p $O
CLIMB_AMD_OPTIONS=$1 rocprofv3 --kernel-trace --output-format csv -d $O -o r -- python $R/bench.py --no-cpu-baseline --no-cls-only-leg --steps 6 --warmup 3 > $O/log 2>&1
python - <<PY
import csv, re, collections
rows=list(csv.DictReader(open("$O/r_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
per=collections.defaultdict(list)
for a,b in zip(idx[2:-1], idx[3:]):
    seq=rows[a+1:b+1]
    names=[re.sub(r"\(.*","",r["Kernel_Name"]).replace("void ","") for r in seq]
    durs=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in seq]
    # label GEMMs by what precedes / follows them
    k=0
    for i,(n,d) in enumerate(zip(names,durs)):
        if "gemm_bf16_nt" in n:
            prev=names[i-1][:22]; nxt=names[i+1][:22] if i+1<len(names) else ""
            per[(n[:44],prev,nxt)].append(d)
        else:
            per[(n[:44],"","")].append(d)
tot=0
for key,v in sorted(per.items(), key=lambda kv:-sum(kv[1])):
    s=sum(v)/ (len(idx)-3)
    tot+=s
    if s>0.02: print(f"{s*1e-3:7.3f} ms/step  n={len(v)/(len(idx)-3):5.1f} avg {sum(v)/len(v):7.1f} us  {key[0]} | after {key[1]} | before {key[2]}")
print("total %.3f ms/step"%(tot*1e-3))
PY
