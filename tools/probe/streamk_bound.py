"""Upper bound on what balancing the 192-tile N = 768 NT GEMMs over all 256 CUs could buy: the same FLOPs laid out as 256 tiles
(M = 16384) with 3/4 of the k-loop.  (A split-tile scheme would add the partial-sum exchange on top of the right-hand numbers.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from climb_amd import _lib
from tools.gemm_bench import timeit, dev, st


def nt(M, N, K, cdt, epi):
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16 if cdt else torch.float32)
    bias = torch.randn(N, device=dev)
    aux = torch.randn(M, N, device=dev) if epi == 2 else None
    return timeit(lambda: _lib.call("climb_gemm_bf16_nt", A, K, W, K, C, N, cdt, M, N, K, bias, epi, aux, N, None, N, None, 0, st())) * 1e6


for rep in range(3):
  for name, K, cdt, epi in [("dhn (bf16 out)", 3072, 1, 0), ("dxn (bf16 out)", 2304, 1, 0), ("down fwd + res (fp32 out)", 3072, 0, 2)]:
    print(f"{name:28s} K={K}: 192 tiles x {K//64} k-steps {nt(12288, 768, K, cdt, epi):6.1f} us | 256 tiles x {K*3//4//64} k-steps {nt(16384, 768, K * 3 // 4, cdt, epi):6.1f} us")
