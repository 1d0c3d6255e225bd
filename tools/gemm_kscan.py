"""K / N scan of the bf16 NT GEMM at M = 12288: separates the k-loop rate (large K asymptote) from per-tile prologue/epilogue
and launch/tail costs (small K).  Run on the GPU box: python tools/gemm_kscan.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from climb_amd import _lib
from tools.gemm_bench import timeit, dev, st

M = 12288
for N in (768, 2304, 3072, 4096):
    for K in (768, 1536, 3072, 6144, 12288):
        A = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: _lib.call("climb_gemm_bf16_nt", A, K, W, K, C, N, 1, M, N, K, None, 0, None, 0, None, 0, None, 0, st()))
        print(f"N={N:5d} K={K:6d}: {t*1e6:8.1f} us  {2.0*M*N*K/t/1e12:7.1f} TF", flush=True)
