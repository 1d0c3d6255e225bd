"""Probe (r03): what the matrix cores sustain as a function of HOW MANY CUs are busy (the library's pipe-only diagnostic, `climb_mfma_sustained_probe`:
16 independent 32 x 32 x 16 accumulations per wave, one wave per SIMD, register operands, no memory traffic), on N(0,1) and on zero operands.
If the total rate saturates before 256 workgroups the chip is at its power budget there, and CUs a GEMM leaves idle (192 tiles of 256 x 192 on
256 CUs: the N = 768 products) come back as clock for the busy ones -- tile quantisation then costs less than the idle fraction suggests.
Second table: the library's persistent NT GEMM on the layer's shapes with N(0,1) against zero operands (same launches, same bytes: the
difference is the clock the power budget allows).  GPU box:  python tools/probe/mfma_power_scan.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from climb_amd import _lib

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
t16 = torch.bfloat16 if _lib.h16() == "bf16" else torch.float16
st = lambda: torch.cuda.current_stream().cuda_stream


def timeit(fn, iters=3, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


out = torch.zeros(256 * 256, device=dev)
iters = 20000
srcs = (("N(0,1)", torch.randn(256 * 256 * 64, device=dev).to(t16)), ("zeros", torch.zeros(256 * 256 * 64, device=dev).to(t16)))
print(f"pipe only ({_lib.h16()} operands), {iters} rounds of 16 MFMAs per wave, 4 waves per workgroup, one workgroup per CU")
print("workgroups |  N(0,1): total TF   per-CU % of 2.5 PF/256 |  zeros: total TF   per-CU %")
for blocks in (32, 64, 96, 128, 160, 192, 224, 256):
    row = []
    for name, src in srcs:
        t = timeit(lambda: _lib.call("climb_mfma_sustained_probe", src, out, blocks, iters, st()))
        fl = blocks * 4 * iters * 16 * 32768.0
        row.append((fl / t / 1e12, fl / t / 1e12 / blocks * 256 / 2500 * 100))
    print(f"   {blocks:4d}    |     {row[0][0]:8.1f}            {row[0][1]:6.1f}          |    {row[1][0]:8.1f}       {row[1][1]:6.1f}")

# the layer's NT products through the library's dispatcher (persistent 256-row tiles), random against zero operands
M = 12288
print("\nNT GEMM C[M,N] = A[M,K] B[N,K]^T, M = 12288, 16-bit output, no epilogue: N(0,1) operands | zero operands")
for name, N, K in (("qkv", 2304, 768), ("up", 3072, 768), ("dhn / down", 768, 3072), ("dctx / out", 768, 768), ("dxn", 768, 2304)):
    res = []
    for kind in ("rand", "zero"):
        A = (torch.randn(M, K, device=dev) if kind == "rand" else torch.zeros(M, K, device=dev)).to(t16)
        B = (torch.randn(N, K, device=dev) if kind == "rand" else torch.zeros(N, K, device=dev)).to(t16)
        C = torch.empty(M, N, device=dev, dtype=t16)
        fn = lambda: _lib.call("climb_gemm_bf16_nt", A, K, B, K, C, N, 1, M, N, K, None, 0, None, 0, None, 0, None, 0, st())
        try:
            t = timeit(fn, iters=20, warm=3)
        except Exception as ex:
            print("skipped:", ex)
            break
        res.append(t)
    if len(res) == 2:
        fl = 2.0 * M * N * K
        print(f"  {name:11s} N={N:5d} K={K:5d}: {res[0]*1e6:7.1f} us {fl/res[0]/1e12:7.1f} TF | {res[1]*1e6:7.1f} us {fl/res[1]/1e12:7.1f} TF | ratio {res[0]/res[1]:.2f}")
