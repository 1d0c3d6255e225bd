"""Probe (r03): is the grouped weight-gradient kernel limited by how well an XCD's 32 concurrent tiles share operand panels?
Homogeneous groups of 48 problems of one shape each (same FLOPs per tile everywhere): the more tiles of a problem, the more panel
sharing inside an XCD's share.  GPU box:  python tools/probe/grouped_dw_locality.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from climb_amd import _lib

dev = torch.device("cuda:0")
M = 12288
st = lambda: torch.cuda.current_stream().cuda_stream


def timeit(fn, iters=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for N, K, count in ((3072, 768, 12), (768, 3072, 12), (2304, 768, 16), (768, 768, 48), (256, 256, 432), (1536, 1536, 12), (3072, 3072, 3)):
    ops = [(torch.randn(M, N, device=dev).bfloat16(), torch.randn(M, K, device=dev).bfloat16(), torch.zeros(N, K, device=dev)) for _ in range(count)]
    rec = np.zeros(count, dtype=[("A", "<u8"), ("B", "<u8"), ("C", "<u8"), ("dbias", "<u8"), ("lda", "<i8"), ("ldb", "<i8"), ("ldc", "<i8"),
                                 ("M", "<i4"), ("N", "<i4"), ("K", "<i4"), ("reserved", "<i4")])
    for r, (dY, X, C) in zip(rec, ops):
        r["A"], r["B"], r["C"], r["lda"], r["ldb"], r["ldc"], r["M"], r["N"], r["K"] = dY.data_ptr(), X.data_ptr(), C.data_ptr(), N, K, K, M, N, K
    Ms, Ns, Ks = (np.ascontiguousarray(rec[f], dtype=np.int32) for f in ("M", "N", "K"))
    tiles = count * (N // 256) * (K // 256)
    cap = tiles + 257
    items, first = np.zeros((cap, 8), dtype=np.int32), np.zeros(257, dtype=np.int32)
    n = _lib.load().climb_tn_grouped_plan(count, Ms.ctypes.data, Ns.ctypes.data, Ks.ctypes.data, 256, items.ctypes.data, cap, first.ctypes.data)
    d = [torch.from_numpy(rec.view(np.uint8).copy()).to(dev), torch.from_numpy(items[:n].copy()).to(dev), torch.from_numpy(first).to(dev)]
    t = timeit(lambda: _lib.call("climb_gemm_bf16_tn_grouped", d[0], d[1], d[2], 256, 0, st()))
    f = 2.0 * M * N * K * count
    operand_mb = count * (N + K) * M * 2 / 1e6
    print(f"{count:3d} x dW[{N:4d} x {K:4d}]: {tiles:4d} tiles ({tiles / 256:.2f} rounds), {t * 1e3:7.3f} ms  {f / t / 1e12:7.1f} TF   operands {operand_mb:7.0f} MB "
          f"-> {operand_mb / 1e6 / t:5.2f} TB/s if read once")
    del ops
    torch.cuda.empty_cache()
