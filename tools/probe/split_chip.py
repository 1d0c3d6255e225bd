"""dX (NT) and dW (TN) of one layer on DISJOINT halves of the chip at the same time (128 persistent workgroups each, two streams) against
the two run back to back on all 256 CUs.  Build: -DTNP_PROBE_GRID=128 for the weight-gradient kernel; option 9 caps the NT grid."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from climb_amd import _lib
from climb_amd.engine import tn_workspace

dev = torch.device("cuda:0")
M = 12288
tn_workspace(dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
half = os.environ.get("HALF", "0") == "1"
if half:
    _lib.call("climb_set_option", 9, 128)


def bench(fn, iters=20):
    with torch.cuda.stream(s1):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s1)
        for _ in range(iters):
            fn()
        e1.record(s1)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, (Nn, Kn, epi), (Nt, Kt) in [("du dgelu (N=3072) || dW2", (3072, 768, 0), (768, 3072)), ("dhn (N=768,K=3072) || dW1", (768, 3072, 0), (3072, 768)),
                                     ("dxn (N=768,K=2304) || dWqkv", (768, 2304, 0), (2304, 768))]:
    A = torch.randn(M, Kn, device=dev).bfloat16(); W = (torch.randn(Nn, Kn, device=dev) * 0.05).bfloat16(); C = torch.empty(M, Nn, device=dev, dtype=torch.bfloat16)
    dY = torch.randn(M, Nt, device=dev).bfloat16(); X = torch.randn(M, Kt, device=dev).bfloat16(); G = torch.zeros(Nt, Kt, device=dev)
    nt = lambda st: _lib.call("climb_gemm_bf16_nt", A, Kn, W, Kn, C, Nn, 1, M, Nn, Kn, None, epi, None, 0, None, 0, None, 0, st)
    tn = lambda st: _lib.call("climb_gemm_bf16_tn", dY, Nt, X, Kt, G, Kt, M, Nt, Kt, None, st)

    def serial():
        nt(s1.cuda_stream); tn(s1.cuda_stream)

    def both():
        s2.wait_stream(s1); nt(s1.cuda_stream); tn(s2.cuda_stream); s1.wait_stream(s2)
    print(f"{'half-chip grids' if half else 'full grids':16s} {name:30s} serial {bench(serial):7.1f} us   concurrent {bench(both):7.1f} us")
