"""K / N scan of the bf16 NT GEMM at M = 12288: separates the k-loop rate (large K asymptote) from per-tile prologue/epilogue
and launch/tail costs (small K).  Run on the GPU box: python tools/gemm_kscan.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from climb_amd import _lib
from tools.gemm_bench import timeit, dev, st

M = 12288
for key, env in ((7, "NT256"), (5, "NT192"), (8, "NT256_PROBE"), (9, "NT256_GRID")):
    if os.environ.get(env) is not None:
        _lib.call("climb_set_option", key, int(os.environ[env]))
NS = [int(v) for v in os.environ.get("NS", "768,2304,3072,4096").split(",")]
KS = [int(v) for v in os.environ.get("KS", "768,1536,3072,6144,12288").split(",")]
for N in NS:
    for K in KS:
        A = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: _lib.call("climb_gemm_bf16_nt", A, K, W, K, C, N, 1, M, N, K, None, 0, None, 0, None, 0, None, 0, st()))
        print(f"N={N:5d} K={K:6d}: {t*1e6:8.1f} us  {2.0*M*N*K/t/1e12:7.1f} TF", flush=True)
