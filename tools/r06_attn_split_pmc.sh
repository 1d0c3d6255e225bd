#!/bin/bash
# MFMA / LDS / wait counters of the split-operand attention kernels inside the bf16x3 step: bash tools/r06_attn_split_pmc.sh   (GPU box)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/prof_attn_split
rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/a -o r -- python $R/bench.py --precision bf16x3 --no-cpu-baseline --no-cls-only-leg --steps 2 --warmup 1 > $O/a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/b -o r -- python $R/bench.py --precision bf16x3 --no-cpu-baseline --no-cls-only-leg --steps 2 --warmup 1 > $O/b.log 2>&1
python - <<PY
import csv, collections, re
def rd(d):
    per = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in csv.DictReader(open("$O/%s/r_counter_collection.csv" % d)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:60]
        per[k][r["Counter_Name"]][r["Dispatch_Id"]] = per[k][r["Counter_Name"]].get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    dur = collections.defaultdict(list)
    for t in csv.DictReader(open("$O/%s/r_kernel_trace.csv" % d)):
        dur[re.sub(r"\(.*", "", t["Kernel_Name"]).replace("void ", "")[:60]].append((int(t["End_Timestamp"]) - int(t["Start_Timestamp"])) / 1e3)
    return per, dur
a, da = rd("a"); b, db = rd("b")
def avg(t, k, c):
    v = list(t.get(k, {}).get(c, {}).values()); return sum(v) / len(v) if v else float("nan")
for k in sorted(a):
    if not any(x in k for x in ("attn", "gemm_bf16_nt4", "tn_grouped", "split_f32", "layernorm")): continue
    d = sum(da[k]) / len(da[k])
    busy, wave = avg(a, k, "SQ_VALU_MFMA_BUSY_CYCLES"), avg(a, k, "SQ_WAVE_CYCLES")
    print(f"{k:62s} {d:8.1f} us  mfma_util {busy / (d * 1e-6 * 2.4e9 * 1024):.3f}  valu_insts {avg(a, k, 'SQ_INSTS_VALU'):.3e}  lds_conflict/active {avg(b, k, 'SQ_LDS_BANK_CONFLICT') / max(1.0, avg(b, k, 'SQ_LDS_IDX_ACTIVE')):.3f}  lds_active {avg(b, k, 'SQ_LDS_IDX_ACTIVE'):.3e}  wait_any/wave {avg(b, k, 'SQ_WAIT_INST_ANY') / max(1.0, wave):.3f} wait_lds/wave {avg(b, k, 'SQ_WAIT_INST_LDS') / max(1.0, wave):.3f}")
PY
