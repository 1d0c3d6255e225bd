"""Micro-benchmark of csrc/heads.hip on the pooler / VQA-head shapes at 64 rows, next to climb_gemm_f32 (split-K allowed) on the same products.
climb_set_option(19, v): 1 = loads only, 2 = MFMAs only (measurement builds of the same kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from climb_amd import _lib

dev = torch.device("cuda:0")
st = lambda: torch.cuda.current_stream().cuda_stream


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


M = 64
shapes = [("pooler", 768, 768, True), ("head.0", 1536, 768, True), ("head.3", 3129, 1536, True), ("d(zn)", 1536, 3129, False), ("d(x)", 768, 1536, False), ("d(clsn)", 768, 768, False)]
for name, N, K, kc in shapes:
    lda = (K + 3) // 4 * 4
    A = torch.randn(M, lda, device=dev)
    W = torch.randn(N, (K + 3) // 4 * 4, device=dev) if kc else torch.randn(K, N, device=dev)
    sbn, sbk = (W.shape[1], 1) if kc else (1, N)
    C = torch.empty(M, N, device=dev)
    bias = torch.randn(N, device=dev)
    row = f"{name:8s} N={N:5d} K={K:5d}"
    for probe in (0, 1, 2):
        _lib.call("climb_set_option", 19, probe)
        t = timeit(lambda: _lib.call("climb_skinny_f32", A, lda, W, sbn, sbk, C, N, M, N, K, bias, 0, None, 0, None, 0.0, None, 0.0, st()))
        row += f"  {('skinny', 'loads only', 'MFMAs only')[probe]} {t:6.1f} us"
    _lib.call("climb_set_option", 19, 0)
    t = timeit(lambda: _lib.call("climb_gemm_f32", A, lda, 1, W, sbn, sbk, C, N, M, N, K, bias, 0, None, 0, None, 0, 0.0, None, 0, 1, st()))
    print(row + f"  gemm_f32 (split-K + zero fill) {t:6.1f} us   [{2.0 * M * N * K / 1e6:.0f} MFLOP]")
