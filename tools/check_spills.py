"""Register-spill report for every kernel of the library (hipcc -Rpass-analysis=kernel-resource-usage).  A kernel that starts to
spill usually still passes its tests and quietly loses half its speed (r01: the fp32 128x128 GEMM, 66 -> 37 TF, when three adapter
epilogue cases were added to its runtime switch).  Usage: python tools/check_spills.py [file.hip ...]; exits non-zero if a kernel
outside the known list uses scratch memory."""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "climb_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# kernels whose spills are a measured trade (see their comments): attention backward phase 1 (168-VGPR cap keeps 3 workgroups per CU),
# the 192x192 GEMM with the fp32 residual / adapter dual-residual operand prefetched (a few dwords, outside the k-loop),
# the fp32 128x128 GEMM's adapter-epilogue instantiation (adapters in the fp32 parity mode only)
# the two-workgroup variant of the persistent GEMM that only option 7 = 4 selects (the persistent 256 x 192 GEMM with the fp32 residual
# epilogue itself no longer spills: r03, the lane id is recomputed per tile instead of kept across the k-loop);
# the grouped weight-gradient launch with the optimizer in its epilogue (r04): 13 dwords of per-lane constants (fragment offsets, lane ids) stored once
# before the item loop and reloaded once per TILE (192 k-tiles each), none inside the k-loop;
# bench.py's pipe-only diagnostic (16 accumulator tiles = all 256 AGPRs + 256 VGPRs at one wave per SIMD: one dword outside its MFMA loop)
KNOWN = ("gemm_bf16_ntsk_kernelIfLi2E", "attn_bwd_bf16_kernelILi1E", "attn_bwd_bf16_fused_kernel", "gemm_bf16_nt192_kernelIfLi2E", "gemm_bf16_nt192_kernelItLi2E", "gemm_bf16_nt192_kernelIfLi7E",
         "gemm_bf16_nt192_kernelItLi7E", "gemm_f32_kernelILi128ELi128ELb1E",
         "gemm_bf16_nt2_kernelIfLi2E", "gemm_bf16_nt2_kernelItLi2E", "mfma_sustained_kernel", "gemm_bf16_tn_grouped_kernelILb0ELb1E")


def report(path):
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Rpass-analysis=kernel-resource-usage", "-c", path, "-o", os.devnull]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    out = []
    name = vgpr = None
    for line in err.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"VGPRs: (\d+)", line)
        if m:
            vgpr = int(m.group(1))
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name:
            out.append((name, vgpr, int(m.group(1))))
            name = None
    return out


def main(files):
    bad = 0
    for f in files:
        for name, vgpr, scratch in report(f):
            if scratch:
                known = any(k in name for k in KNOWN)
                print(f"{'known' if known else 'SPILL'}  {os.path.basename(f)}  {name}  VGPRs {vgpr}  scratch {scratch} B/lane")
                bad += 0 if known else 1
    print("no unexpected spills" if not bad else f"{bad} kernel(s) spill unexpectedly")
    return bad


if __name__ == "__main__":
    fs = sys.argv[1:] or [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    sys.exit(1 if main(fs) else 0)
