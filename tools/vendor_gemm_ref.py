"""Reference point, not product code: what the vendor GEMM library (hipBLASLt / rocBLAS behind torch.mm on ROCm) reaches on the plain products of a
layer -- the NT shapes that carry no fused epilogue (or a bias only) -- next to this repository's hand-written kernels, same operands, settings
interleaved, outputs rotated over 6 buffers (HBM-resident, as in a step).  The product never calls the library (README: hand-written gfx950 kernels);
this exists so that `roofline.frac` can be read against what the chip's own library sustains on these shapes.   GPU box: python tools/vendor_gemm_ref.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from climb_amd import _lib

dev = torch.device("cuda:0")
st = lambda: torch.cuda.current_stream().cuda_stream
M = 12288


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


tot = {"ours": 0.0, "vendor": 0.0}
for name, N, K, bias in [("qkv fwd (bias)", 2304, 768, True), ("dhn", 768, 3072, False), ("dctx", 768, 768, False), ("dxn", 768, 2304, False),
                         ("up fwd, no gelu", 3072, 768, True), ("down fwd, no residual", 768, 3072, True)]:
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    b = torch.randn(N, device=dev)
    b16 = b.bfloat16()
    Cs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(6)]
    ctr = [0]

    def ours():
        ctr[0] += 1
        _lib.call("climb_gemm_bf16_nt", A, K, W, K, Cs[ctr[0] % 6], N, 1, M, N, K, b if bias else None, 0, None, 0, None, 0, None, 0, st())

    Wt = W.t()

    def vendor():
        ctr[0] += 1
        if bias:
            torch.addmm(b16, A, Wt, out=Cs[ctr[0] % 6])
        else:
            torch.mm(A, Wt, out=Cs[ctr[0] % 6])
    ts = {"ours": [], "vendor": []}
    for _ in range(5):
        ts["ours"].append(timeit(ours))
        ts["vendor"].append(timeit(vendor))
    med = {k: sorted(v)[2] for k, v in ts.items()}
    f = 2.0 * M * N * K
    for k in med:
        tot[k] += med[k]
    print(f"{name:24s} N={N:5d} K={K:5d}: ours {med['ours']:7.1f} us {f / med['ours'] / 1e6:7.1f} TF | vendor library {med['vendor']:7.1f} us {f / med['vendor'] / 1e6:7.1f} TF", flush=True)
print(f"sum: ours {tot['ours']:.1f} us | vendor library {tot['vendor']:.1f} us")
