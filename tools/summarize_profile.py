"""gpurun_out/prof_<tag>/ (tools/profile_step.sh) -> profiles/<tag>_*: per-kernel time, HBM-side traffic (FETCH_SIZE doubled: the gfx950
correction of MI355X_MICROARCH.md section HBM), MFMA-busy estimate, and <tag>_traffic.json keyed by the hash of the kernel sources.
    python tools/summarize_profile.py r02"""
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"          # r06: `python tools/summarize_profile.py r06 bf16x3` summarises the split-operand mode's passes
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}" + ("" if prec == "bf16" else f"_{prec}"))
dst = os.path.join(ROOT, "profiles")
out_tag = tag if prec == "bf16" else f"{tag}_{prec}"          # (only the bf16 passes write rNN_traffic.json, the file bench.py's `roofline.traffic` reads)


def short(name):
    return re.sub(r"\(.*", "", name).strip()


def read_counters(d):
    """kernel -> counter -> [values per dispatch]; and kernel -> [durations us]"""
    rows = list(csv.DictReader(open(os.path.join(src, d, "r_counter_collection.csv"))))
    trace = {r["Dispatch_Id"]: r for r in csv.DictReader(open(os.path.join(src, d, "r_kernel_trace.csv")))}
    per = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in rows:
        per[short(r["Kernel_Name"])][r["Counter_Name"]][r["Dispatch_Id"]] = per[short(r["Kernel_Name"])][r["Counter_Name"]].get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    dur = collections.defaultdict(list)
    for t in trace.values():
        dur[short(t["Kernel_Name"])].append((int(t["End_Timestamp"]) - int(t["Start_Timestamp"])) / 1e3)
    return per, dur


# 1. kernel stats, per STEP: the trace holds process start-up too (parameter uploads = hundreds of copyBuffer launches, allocator fills); a step
# is delimited by its one grouped weight-gradient launch, and only launches between the first and the last delimiter are counted and divided
# by the number of whole steps between them (VERDICT r4 next #3 iv: start-up copies are not part of a step)
def step_window(trace_rows, skip=("mfma_sustained_kernel",)):
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in trace_rows), key=lambda t: t[0])
    marks = [t0 for t0, _, k in rows if "gemm_bf16_tn_grouped_kernel" in k]
    if len(marks) < 2:
        raise SystemExit("fewer than two grouped weight-gradient launches in the trace: cannot delimit steps")
    inside = [(t0, t1, k) for t0, t1, k in rows if marks[0] <= t0 < marks[-1] and not any(x in k for x in skip)]
    return inside, len(marks) - 1


trace_rows = list(csv.DictReader(open(os.path.join(src, "stats", "r_kernel_trace.csv"))))
inside, steps = step_window(trace_rows)
agg = collections.OrderedDict()
for t0, t1, k in inside:
    a = agg.setdefault(k, [0, 0])
    a[0] += 1
    a[1] += t1 - t0
tot_ns = sum(a[1] for a in agg.values())
with open(os.path.join(dst, f"{tag}_{prec}_bs64_kernel_stats.csv"), "w") as f:
    f.write("kernel,calls_per_step,avg_us,total_ms_per_step,percent\n")
    for k, (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write(f"\"{k}\",{n / steps:.1f},{ns / n / 1e3:.2f},{ns / steps / 1e6:.4f},{100.0 * ns / tot_ns:.2f}\n")
total_ms = tot_ns / steps / 1e6
launches_per_step = sum(a[0] for a in agg.values()) / steps
# 2. traffic (same windowing: whole steps of each PMC pass)
def read_traffic(d, counter):
    """kernel -> [per-dispatch counter value (summed over its instances)] for dispatches inside the pass's whole steps; and the number of steps"""
    tr = list(csv.DictReader(open(os.path.join(src, d, "r_kernel_trace.csv"))))
    rows = sorted(((int(r["Start_Timestamp"]), r["Dispatch_Id"], short(r["Kernel_Name"])) for r in tr), key=lambda t: t[0])
    marks = [t0 for t0, _, k in rows if "gemm_bf16_tn_grouped_kernel" in k]
    keep = {d_ for t0, d_, k in rows if marks[0] <= t0 < marks[-1]}
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(os.path.join(src, d, "r_counter_collection.csv"))):
        if r["Counter_Name"] == counter and r["Dispatch_Id"] in keep:
            per[short(r["Kernel_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: list(v.values()) for k, v in per.items()}, len(marks) - 1


fetch, steps_f = read_traffic("pmc_FETCH_SIZE", "FETCH_SIZE")
write, steps_w = read_traffic("pmc_WRITE_SIZE", "WRITE_SIZE")
rows = []
for k in sorted(set(fetch) | set(write)):
    if "mfma_sustained_kernel" in k:
        continue
    fv, wv = fetch.get(k, []), write.get(k, [])
    if not fv or not wv:
        continue
    f_kb, w_kb = sum(fv) / len(fv), sum(wv) / len(wv)
    rows.append((k, len(fv) / steps_f, f_kb, w_kb, int((2 * f_kb + w_kb) * 1024)))
rows.sort(key=lambda r: -r[4] * r[1])
with open(os.path.join(dst, f"{out_tag}_hbm_traffic_per_kernel.csv"), "w") as f:
    f.write("kernel,launches_per_step,FETCH_SIZE_KB_avg,WRITE_SIZE_KB_avg,hbm_bytes_per_launch_corrected,hbm_GB_per_step\n")
    for k, n, a, b, c in rows:
        f.write(f"\"{k}\",{n:.1f},{a:.1f},{b:.1f},{c},{c * n / 1e9:.3f}\n")
step_bytes = sum(c * n for _, n, _, _, c in rows)
nt = [(n, c) for k, n, _, _, c in rows if "gemm_bf16_nt" in k]
nt_avg = sum(n * c for n, c in nt) / max(1e-9, sum(n for n, _ in nt))
import bench
json.dump({"csrc_sha16": bench.csrc_hash(), "step_sha16": bench.step_hash(),
           "gemm_bf16_nt": {"hbm_bytes_per_launch": int(nt_avg), "launches_per_step": round(sum(n for n, _ in nt), 1),
           "source": f"profiles/{out_tag}_hbm_traffic_per_kernel.csv: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, FETCH_SIZE doubled (gfx950 "
                     "correction), averaged over every bf16 NT launch (all tile variants) of the passes' whole steps"},
           "hbm_bytes_per_step": int(step_bytes), "kernel_ms_per_step": round(total_ms, 3), "launches_per_step": round(launches_per_step, 1),
           "steps_in_window": {"stats": steps, "fetch": steps_f, "write": steps_w}},
          open(os.path.join(dst, f"{tag}_traffic.json" if prec == "bf16" else f"{tag}_traffic_{prec}.json"), "w"), indent=1)
# 3. MFMA / LDS counters for the GEMM kernels
mf, dur = read_counters("pmc_mfma")
ld, _ = read_counters("pmc_lds")
with open(os.path.join(dst, f"{out_tag}_gemm_bf16_pmc_summary.csv"), "w") as f:
    f.write("kernel,launches,avg_duration_us_under_pmc,SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CYCLES,SQ_WAVE_CYCLES,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_WAIT_INST_ANY,mfma_util_est\n")
    for k in sorted(mf):
        if "gemm_bf16" not in k and "attn" not in k and "tn_grouped" not in k:
            continue
        def avg(tab, c):
            v = list(tab.get(k, {}).get(c, {}).values())
            return sum(v) / len(v) if v else float("nan")
        d = sum(dur[k]) / len(dur[k])
        busy = avg(mf, "SQ_VALU_MFMA_BUSY_CYCLES")
        util = busy / (d * 1e-6 * 2.4e9 * 1024)          # MFMA-busy cycles / (launch duration x 2.4 GHz x 1024 SIMDs)
        f.write(f"\"{k}\",{len(dur[k])},{d:.1f},{busy:.4e},{avg(mf, 'SQ_BUSY_CYCLES'):.4e},{avg(mf, 'SQ_WAVE_CYCLES'):.4e},{avg(ld, 'SQ_LDS_BANK_CONFLICT'):.4e},"
                f"{avg(ld, 'SQ_LDS_IDX_ACTIVE'):.4e},{avg(ld, 'SQ_WAIT_INST_ANY'):.4e},{util:.3f}\n")
print(f"kernel time {total_ms:.3f} ms/step; HBM-side traffic {step_bytes / 1e9:.2f} GB/step; NT avg {nt_avg / 1e6:.1f} MB/launch")
