"""Probe (r03): weight-gradient k-loop with one wave per SIMD and 128 x 128 wave tiles (tools/probe/csrc/tn_wave128.hip) against the library's
8-wave kernels on the same problem: C[N,K] = A[M,N]^T B[M,K], M = 12288 tokens, 256 output tiles of 256 x 256 = one round of the chip.
Build here (see the .hip header), run on the GPU box:  python tools/probe/tn_wave128.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from climb_amd import _lib

dev = torch.device("cuda:0")
_here = os.path.dirname(os.path.abspath(__file__))
_so = os.path.join(_here, "csrc", "libtn_wave128.so")
if not os.path.exists(_so) or os.path.getmtime(_so) < os.path.getmtime(os.path.join(_here, "csrc", "tn_wave128.hip")):
    import subprocess
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-I", os.path.join(_here, "..", "..", "climb_amd", "csrc"),
                           os.path.join(_here, "csrc", "tn_wave128.hip"), "-o", _so])
lib = ctypes.CDLL(_so)
lib.probe_tn_wave128_s32i.argtypes = lib.probe_tn_wave128_s32.argtypes = lib.probe_tn_wave128.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_int, ctypes.c_void_p]


def probe(A, B, C, fn=None):
    M, N = A.shape
    K = B.shape[1]
    rc = (fn or lib.probe_tn_wave128)(A.data_ptr(), N, B.data_ptr(), K, C.data_ptr(), K, M, N, K, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


g = torch.Generator(device="cpu").manual_seed(0)
for M, N, K in ((256, 256, 256), (1024, 512, 768)):
    A = torch.randn(M, N, generator=g).bfloat16().to(dev)
    B = torch.randn(M, K, generator=g).bfloat16().to(dev)
    C = torch.full((N, K), float("nan"), device=dev)
    ref = A.float().t() @ B.float()
    for name, fn in (("64-token stages x 2", lib.probe_tn_wave128), ("32-token stages x 4", lib.probe_tn_wave128_s32), ("32-token stages x 4, interleaved", lib.probe_tn_wave128_s32i)):
        C.fill_(float("nan"))
        probe(A, B, C, fn)
        err = float((C - ref).abs().max() / ref.abs().max())
        print(f"correctness {name} M={M} N={N} K={K}: max rel err {err:.2e}")
        assert err < 1e-5, err

M = 12288
for N, K in ((4096, 4096), (2048, 4096), (768, 3072), (3072, 768)):
    A = torch.randn(M, N, device=dev).bfloat16()
    B = torch.randn(M, K, device=dev).bfloat16()
    C = torch.zeros(N, K, device=dev)
    flops = 2.0 * M * N * K
    tiles = (N // 256) * (K // 256)
    t = timeit(lambda: probe(A, B, C))
    print(f"probe   N={N:5d} K={K:5d} ({tiles:3d} tiles): {t*1e6:8.1f} us  {flops/t/1e12:7.1f} TF   {t*1e6/(M/64):6.3f} us per 64 tokens  (2 stages of 64 tokens)")
    t = timeit(lambda: probe(A, B, C, lib.probe_tn_wave128_s32))
    print(f"probe   N={N:5d} K={K:5d} ({tiles:3d} tiles): {t*1e6:8.1f} us  {flops/t/1e12:7.1f} TF   {t*1e6/(M/64):6.3f} us per 64 tokens  (4 stages of 32 tokens, 3 in flight)")
    t = timeit(lambda: probe(A, B, C, lib.probe_tn_wave128_s32i))
    print(f"probe   N={N:5d} K={K:5d} ({tiles:3d} tiles): {t*1e6:8.1f} us  {flops/t/1e12:7.1f} TF   {t*1e6/(M/64):6.3f} us per 64 tokens  (the same, reads / DMA between the MFMAs)")
    t = timeit(lambda: _lib.call("climb_gemm_bf16_tn", A, N, B, K, C, K, M, N, K, None, torch.cuda.current_stream().cuda_stream))
    print(f"library N={N:5d} K={K:5d} (climb_gemm_bf16_tn, its own split choice): {t*1e6:8.1f} us  {flops/t/1e12:7.1f} TF")

# DVFS check (microarchitecture guide: the chip clocks to its power budget, zero operands toggle less): the single-round problem on zeros
N = K = 4096
A = torch.zeros(M, N, device=dev).bfloat16()
B = torch.zeros(M, K, device=dev).bfloat16()
C = torch.zeros(N, K, device=dev)
flops = 2.0 * M * N * K
for name, fn in (("2 x 64-token stages", lib.probe_tn_wave128), ("interleaved", lib.probe_tn_wave128_s32i)):
    t = timeit(lambda: probe(A, B, C, fn))
    print(f"zeros   N={N:5d} K={K:5d} (256 tiles, {name}): {t*1e6:8.1f} us  {flops/t/1e12:7.1f} TF   {t*1e6/(M/64):6.3f} us per 64 tokens")
t = timeit(lambda: _lib.call("climb_gemm_bf16_tn", A, N, B, K, C, K, M, N, K, None, torch.cuda.current_stream().cuda_stream))
print(f"zeros   library N={N:5d} K={K:5d}: {t*1e6:8.1f} us  {flops/t/1e12:7.1f} TF")

# the MFMA pipe alone (register operands, 16 independent accumulators per wave, one wave per SIMD) on all 256 CUs / on 32: what the chip sustains
lib.probe_mfma_only.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
out = torch.zeros(256 * 256, device=dev)
iters = 20000
for blocks in (256, 32):
    for name, src in (("random N(0,1)", torch.randn(256 * 256 * 64, device=dev).bfloat16()), ("zeros", torch.zeros(256 * 256 * 64, device=dev).bfloat16())):
        fn = lambda: lib.probe_mfma_only(src.data_ptr(), out.data_ptr(), blocks, iters, torch.cuda.current_stream().cuda_stream)
        t = timeit(fn, iters=3)
        fl = blocks * 4 * iters * 16 * 32768.0
        print(f"MFMA only, {blocks:3d} workgroups, {name:14s}: {fl/t/1e12:7.1f} TF  ({t/(iters*16)*1e9:5.2f} ns per MFMA per SIMD = {fl/t/1e12/blocks*256/2500*100:5.1f} % of 2.5 PF per-CU rate)")
