"""Probe (r03): does de-synchronising the workgroups' epilogue bursts pay?  The up-projection (M = 12288, N = 3072, K = 768, GELU + saved
pre-activation: 151 MB of stores) as TWO concurrent launches on two streams -- columns [0, 1536) in 256 x 192 tiles on 128 CUs and
columns [1536, 3072) in 256 x 256 tiles on the other 128 CUs (3 x 13 us vs 2.25 x 17 us of k-loop per CU: balanced, but the two
halves reach their epilogues at different times) -- against the one launch of 768 equal tiles in lock step.
GPU box:  python tools/probe/hetero_tiles.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from climb_amd import _lib

dev = torch.device("cuda:0")
M, N, K = 12288, 3072, 768
A = torch.randn(M, K, device=dev).bfloat16()
W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
bias = torch.randn(N, device=dev)
C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
U = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def whole():
    _lib.call("climb_set_option", 7, 1)
    _lib.call("climb_set_option", 9, 256)
    _lib.call("climb_gemm_bf16_nt", A, K, W, K, C, N, 1, M, N, K, bias, 1, None, N, U, N, None, 0, torch.cuda.current_stream().cuda_stream)


def halves(split=1536, wa=3, wb=2, ga=128, gb=128):
    cur = torch.cuda.current_stream()
    ev = torch.cuda.Event()
    ev.record(cur)
    done = []
    for st, lo, hi, width, grid in ((s1, 0, split, wa, ga), (s2, split, N, wb, gb)):
        st.wait_event(ev)
        _lib.call("climb_set_option", 7, width)          # 3: force 192 columns, 2: force 256
        _lib.call("climb_set_option", 9, grid)
        n = hi - lo
        _lib.call("climb_gemm_bf16_nt", A, K, W[lo:], K, C[:, lo:], N, 1, M, n, K, bias[lo:], 1, None, N, U[:, lo:], N, None, 0, st.cuda_stream)
        e = torch.cuda.Event()
        e.record(st)
        done.append(e)
    for e in done:
        cur.wait_event(e)


def timeit(fn, iters=20):
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


whole()
ref = C.clone()
halves()
torch.cuda.synchronize()
print("halves == whole:", bool(torch.equal(ref, C)))
print(f"one launch, 768 tiles of 256 x 192 in lock step: {timeit(whole):7.1f} us")
for kw in (dict(), dict(wa=3, wb=3), dict(wa=2, wb=2), dict(split=1344, wa=3, wb=2), dict(split=1536, wa=3, wb=2, ga=112, gb=144)):
    print(f"two concurrent launches {kw or '(192-wide | 256-wide, 128 CUs each)'}: {timeit(lambda: halves(**kw)):7.1f} us")
_lib.call("climb_set_option", 7, 1)
_lib.call("climb_set_option", 9, 256)
