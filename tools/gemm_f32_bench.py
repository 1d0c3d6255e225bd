"""fp32 (parity-mode) GEMM at the per-layer shapes: exact-f32 MFMA rate.  Run on the GPU box: python tools/gemm_f32_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from climb_amd import _lib
from tools.gemm_bench import timeit, dev, st

M = 12288
for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.05
    C = torch.empty(M, N, device=dev)
    b = torch.randn(N, device=dev)
    t = timeit(lambda: _lib.call("climb_gemm_f32", A, K, 1, W, K, 1, C, N, M, N, K, b, 0, None, 0, None, 0, 0.0, None, 0, 0, st()), iters=5)
    print(f"f32 NT N={N:5d} K={K:5d}: {t*1e6:8.1f} us  {2.0*M*N*K/t/1e12:6.1f} TF", flush=True)
    dY = torch.randn(M, N, device=dev)
    G = torch.zeros(N, K, device=dev)
    t = timeit(lambda: _lib.call("climb_gemm_f32", dY, 1, N, A, 1, K, G, K, N, K, M, None, 0, None, 0, None, 0, 1.0, None, 0, 0, st()), iters=5)
    print(f"f32 TN N={N:5d} K={K:5d}: {t*1e6:8.1f} us  {2.0*M*N*K/t/1e12:6.1f} TF", flush=True)
