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
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")


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


# 1. kernel stats
stats = list(csv.DictReader(open(os.path.join(src, "stats", "r_kernel_stats.csv"))))
# bench.py's pipe-only diagnostic (two launches AFTER the timed region: roofline.sustained_mfma_tflops_random_operands) is not part of a step
stats = [r for r in stats if "mfma_sustained_kernel" not in r["Name"]]
tot_ns = sum(float(r["TotalDurationNs"]) for r in stats)
steps = 11
with open(os.path.join(dst, f"{tag}_bf16_bs64_kernel_stats.csv"), "w") as f:
    f.write("kernel,calls_per_step,avg_us,total_ms_per_step,percent\n")
    for r in stats:
        f.write(f"\"{short(r['Name'])}\",{int(r['Calls']) / steps:.1f},{float(r['AverageNs']) / 1e3:.2f},{float(r['TotalDurationNs']) / steps / 1e6:.4f},{100.0 * float(r['TotalDurationNs']) / tot_ns:.2f}\n")
total_ms = sum(float(r["TotalDurationNs"]) for r in stats) / steps / 1e6
# 2. traffic
fetch, _ = read_counters("pmc_FETCH_SIZE")
write, _ = read_counters("pmc_WRITE_SIZE")
rows = []
for k in sorted(set(fetch) | set(write)):
    if "mfma_sustained_kernel" in k:
        continue
    fv = list(fetch.get(k, {}).get("FETCH_SIZE", {}).values())
    wv = list(write.get(k, {}).get("WRITE_SIZE", {}).values())
    if not fv or not wv:
        continue
    f_kb, w_kb = sum(fv) / len(fv), sum(wv) / len(wv)
    rows.append((k, len(fv), f_kb, w_kb, int((2 * f_kb + w_kb) * 1024)))
rows.sort(key=lambda r: -r[4] * r[1])
with open(os.path.join(dst, f"{tag}_hbm_traffic_per_kernel.csv"), "w") as f:
    f.write("kernel,launches_in_trace,FETCH_SIZE_KB_avg,WRITE_SIZE_KB_avg,hbm_bytes_per_launch_corrected\n")
    for k, n, a, b, c in rows:
        f.write(f"\"{k}\",{n},{a:.1f},{b:.1f},{c}\n")
steps_pmc = 3
step_bytes = sum(c * n for _, n, _, _, c in rows) / steps_pmc
nt = [(n, c) for k, n, _, _, c in rows if "gemm_bf16_nt" in k]
nt_avg = sum(n * c for n, c in nt) / max(1, sum(n for n, _ in nt))
import bench
json.dump({"csrc_sha16": bench.csrc_hash(), "gemm_bf16_nt": {"hbm_bytes_per_launch": int(nt_avg), "launches_averaged": sum(n for n, _ in nt),
           "source": f"profiles/{tag}_hbm_traffic_per_kernel.csv: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, FETCH_SIZE doubled (gfx950 "
                     "correction), averaged over every bf16 NT launch (all tile variants) of 3 steps"},
           "hbm_bytes_per_step": int(step_bytes), "kernel_ms_per_step": round(total_ms, 3)}, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
# 3. MFMA / LDS counters for the GEMM kernels
mf, dur = read_counters("pmc_mfma")
ld, _ = read_counters("pmc_lds")
with open(os.path.join(dst, f"{tag}_gemm_bf16_pmc_summary.csv"), "w") as f:
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
