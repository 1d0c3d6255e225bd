#!/bin/bash
# r05: SQ counters of the four-wave NT kernel's measurement builds (bash tools/nt4_anatomy_pmc.sh on the GPU box; PMC in its own pass, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/prof_nt4_anatomy
rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O -o r -- python $R/tools/nt4_check.py --quick --no-bench --anatomy2 > $O/log 2>&1
python - <<PY
import csv, re, collections
rows=list(csv.DictReader(open("$O/r_counter_collection.csv")))
tr={r["Dispatch_Id"]:r for r in csv.DictReader(open("$O/r_kernel_trace.csv"))}
per=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter(); dur=collections.defaultdict(float)
seen=set()
for r in rows:
    if "gemm_bf16_nt4" not in r["Kernel_Name"]: continue
    t=tr[r["Dispatch_Id"]]
    k=re.sub(r"\(.*","",r["Kernel_Name"]).replace("void ","")[:44]+" grid"+t["Grid_Size_X"]
    per[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); cnt[k]+=1
        dur[k]+=(int(t["End_Timestamp"])-int(t["Start_Timestamp"]))/1e3
for k in sorted(per):
    n=cnt[k]; c=per[k]
    wc=c["SQ_WAVE_CYCLES"]
    print(f"{k:56s} n={n:4d} {dur[k]/n:7.1f} us  clock {c['GRBM_GUI_ACTIVE']/n/ (dur[k]/n*1e-6)/1e9:5.2f} GHz  wave-cycles/wave {4*wc/n/1024:9.0f}  mfma_busy {c['SQ_VALU_MFMA_BUSY_CYCLES']/(4*wc):5.3f}  wait_any {c['SQ_WAIT_ANY']/wc:5.3f} wait_inst {c['SQ_WAIT_INST_ANY']/wc:5.3f} (lds {c['SQ_WAIT_INST_LDS']/wc:5.3f}) active {c['SQ_ACTIVE_INST_ANY']/wc:5.3f}")
PY
