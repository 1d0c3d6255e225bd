"""Does a weight-gradient GEMM running on a second stream fill the 64 CUs the 192-tile N = 768 NT GEMMs leave idle?
Pairs of the backward that share their input (dX and dW of one linear layer): serial on one stream vs concurrent on two."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from climb_amd import _lib
from climb_amd.engine import tn_workspace

dev = torch.device("cuda:0")
M = 12288
tn_workspace(dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(pairs, concurrent, iters=20):
    def once():
        for nt, tn in pairs:
            if concurrent:
                s2.wait_stream(s1)
                nt(s1.cuda_stream)
                tn(s2.cuda_stream)
                s1.wait_stream(s2)
            else:
                nt(s1.cuda_stream)
                tn(s1.cuda_stream)
    with torch.cuda.stream(s1):
        for _ in range(3):
            once()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s1)
        for _ in range(iters):
            once()
        e1.record(s1)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def make(Nn, Kn, Nt, Kt):
    # NT: dX[M, Nn] = dY[M, Kn] W^T ; TN: dW[Nt, Kt] = dY[M, Nt]^T X[M, Kt]
    A = torch.randn(M, Kn, device=dev).bfloat16(); W = (torch.randn(Nn, Kn, device=dev) * 0.05).bfloat16(); C = torch.empty(M, Nn, device=dev, dtype=torch.bfloat16)
    dY = A if Kn == Nt else torch.randn(M, Nt, device=dev).bfloat16(); X = torch.randn(M, Kt, device=dev).bfloat16(); G = torch.zeros(Nt, Kt, device=dev)
    nt = lambda st: _lib.call("climb_gemm_bf16_nt", A, Kn, W, Kn, C, Nn, 1, M, Nn, Kn, None, 0, None, 0, None, 0, None, 0, st)
    tn = lambda st: _lib.call("climb_gemm_bf16_tn", dY, Nt, X, Kt, G, Kt, M, Nt, Kt, None, st)
    return nt, tn


for name, shp in [("dhn || dW1", (768, 3072, 3072, 768)), ("dxn || dWqkv", (768, 2304, 2304, 768)), ("dctx || dWo", (768, 768, 768, 768))]:
    p = make(*shp)
    print(f"{name:14s} serial {run([p], False):7.1f} us   concurrent {run([p], True):7.1f} us")
