"""Can the split-sum reduce of a weight-gradient GEMM hide under the NEXT GEMM on a second stream?  (probe build: -DTNP_PROBE_REDUCE_ENTRY)"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from climb_amd import _lib

dev = torch.device("cuda:0")
M = 12288
lib = ctypes.CDLL(os.environ["CLIMB_AMD_LIB"])
lib.climb_probe_tn_reduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


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


for name, (Nn, Kn, cdt, epi), (Nt, Kt, splits) in [("dW2 reduce || dhn", (768, 3072, 1, 0), (768, 3072, 5)), ("dW1 reduce || LN-sized stream", None, (3072, 768, 5)),
                                                   ("dWqkv reduce || dxn... (qkv NT fwd shape)", (2304, 768, 1, 0), (2304, 768, 7)), ("dWo reduce || dctx", (768, 768, 1, 0), (768, 768, 21))]:
    slab = torch.randn(splits * Nt * Kt, device=dev); C = torch.zeros(Nt, Kt, device=dev)
    red = lambda st: lib.climb_probe_tn_reduce(slab.data_ptr(), C.data_ptr(), Kt, Nt, Kt, splits, st)
    if Nn is None if False else name.endswith("stream"):
        x = torch.randn(M, 768, device=dev); y = torch.empty_like(x)
        other = lambda st: y.copy_(x)          # 75 MB of HBM traffic on the current stream
        with torch.cuda.stream(s1):
            pass
        nt = lambda st: other(st)
    else:
        A = torch.randn(M, Kn, device=dev).bfloat16(); W = (torch.randn(Nn, Kn, device=dev) * 0.05).bfloat16(); Cn = torch.empty(M, Nn, device=dev, dtype=torch.bfloat16)
        nt = lambda st: _lib.call("climb_gemm_bf16_nt", A, Kn, W, Kn, Cn, Nn, 1, M, Nn, Kn, None, 0, None, 0, None, 0, None, 0, st)

    def serial():
        red(s1.cuda_stream); nt(s1.cuda_stream)

    def overlapped():
        s2.wait_stream(s1)
        red(s2.cuda_stream)
        nt(s1.cuda_stream)
        s1.wait_stream(s2)
    t_r = bench(lambda: red(s1.cuda_stream)); t_n = bench(lambda: nt(s1.cuda_stream))
    print(f"{name:44s} reduce {t_r:6.1f}  other {t_n:6.1f}  serial {bench(serial):6.1f}  overlapped {bench(overlapped):6.1f} us")
