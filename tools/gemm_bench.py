"""Micro-benchmark of the bf16 GEMM shapes of one ViLT step (bs=64: M = 12288 token rows).  Run on the GPU box:
    python tools/gemm_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from climb_amd import _lib

dev = torch.device("cuda:0")
M = int(os.environ.get("M", 12288))
st = lambda: torch.cuda.current_stream().cuda_stream


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
    return e0.elapsed_time(e1) / iters * 1e-3


if __name__ == "__main__":
    if os.environ.get("NT_WAVES") is not None:
        _lib.call("climb_set_option", 1, int(os.environ["NT_WAVES"]))
    if os.environ.get("NT256") is not None:
        _lib.call("climb_set_option", 7, int(os.environ["NT256"]))
    if os.environ.get("NT256_GRID") is not None:
        _lib.call("climb_set_option", 9, int(os.environ["NT256_GRID"]))
    if os.environ.get("DEPHASE") is not None:
        _lib.call("climb_set_option", 11, int(os.environ["DEPHASE"]))
    if os.environ.get("TNP") is not None:
        _lib.call("climb_set_option", 10, int(os.environ["TNP"]))
    if os.environ.get("NT192") is not None:
        _lib.call("climb_set_option", 5, int(os.environ["NT192"]))
    if os.environ.get("TN_WAVES") is not None:
        _lib.call("climb_set_option", 6, int(os.environ["TN_WAVES"]))
    if os.environ.get("NT96") is not None:
        _lib.call("climb_set_option", 4, int(os.environ["NT96"]))
    if os.environ.get("TN_TARGET") is not None:
        _lib.call("climb_set_option", 3, int(os.environ["TN_TARGET"]))
    if os.environ.get("NT_SMALL_M") is not None:
        _lib.call("climb_set_option", 2, int(os.environ["NT_SMALL_M"]))
    if os.environ.get("TN_WS", "1") != "0":
        from climb_amd.engine import tn_workspace
        tn_workspace(dev)
    if os.environ.get("NT_WS", "0") != "0":          # r03: scratch of the split-along-K NT kernel (CLIMB_AMD_OPTIONS=14=1 NT_WS=1)
        from climb_amd.engine import nt_workspace
        nt_workspace(dev)
    tot_t = tot_f = 0.0
    for name, N, K, cdt, epi in [("qkv fwd", 2304, 768, 1, 0), ("out fwd +res", 768, 768, 0, 2), ("up fwd gelu", 3072, 768, 1, 1), ("down fwd +res", 768, 3072, 0, 2),
                                 ("du dgelu", 3072, 768, 1, 3), ("dhn", 768, 3072, 1, 0), ("dctx", 768, 768, 1, 0), ("dxn", 768, 2304, 1, 0)]:
        A = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16 if cdt else torch.float32)
        bias = torch.randn(N, device=dev)
        aux = torch.randn(M, N, device=dev) if epi == 2 else (torch.randn(M, N, device=dev).bfloat16() if epi == 3 else None)
        auxo = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi == 1 else None
        t = timeit(lambda: _lib.call("climb_gemm_bf16_nt", A, K, W, K, C, N, cdt, M, N, K, bias, epi, aux, N, auxo, N, None, 0, st()))
        f = 2.0 * M * N * K
        tot_t += t; tot_f += f
        print(f"NT {name:14s} N={N:5d} K={K:5d}: {t*1e6:8.1f} us  {f/t/1e12:7.1f} TF")
    print(f"NT total {tot_t*1e3:.3f} ms/layer  {tot_f/tot_t/1e12:.1f} TF")
    tt = tf = 0.0
    for name, N, K in [("dW2", 768, 3072), ("dW1", 3072, 768), ("dWo", 768, 768), ("dWqkv", 2304, 768)]:
        dY = torch.randn(M, N, device=dev).bfloat16()
        X = torch.randn(M, K, device=dev).bfloat16()
        C = torch.zeros(N, K, device=dev)
        t = timeit(lambda: _lib.call("climb_gemm_bf16_tn", dY, N, X, K, C, K, M, N, K, None, st()))
        f = 2.0 * M * N * K
        tt += t; tf += f
        print(f"TN {name:14s} N={N:5d} K={K:5d}: {t*1e6:8.1f} us  {f/t/1e12:7.1f} TF")
    print(f"TN total {tt*1e3:.3f} ms/layer  {tf/tt/1e12:.1f} TF")
    # r03: the same weight gradients as grouped launches (one launch per `layers` layers; per-layer operands like the engine keeps them)
    import numpy as np
    L = int(os.environ.get("GROUP_LAYERS_MAX", 12))
    shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
    ops = [[(torch.randn(M, N, device=dev).bfloat16(), torch.randn(M, K, device=dev).bfloat16(), torch.zeros(N, K, device=dev)) for N, K in shapes] for _ in range(L)]
    for layers in [g for g in (12, 6, 4, 3, 2, 1) if g <= L]:
        flat = [o for lay in ops[:layers] for o in lay]
        rec = np.zeros(len(flat), dtype=[("A", "<u8"), ("B", "<u8"), ("C", "<u8"), ("dbias", "<u8"), ("lda", "<i8"), ("ldb", "<i8"), ("ldc", "<i8"),
                                         ("M", "<i4"), ("N", "<i4"), ("K", "<i4"), ("reserved", "<i4")])
        for r, (dY, X, C) in zip(rec, flat):
            r["A"], r["B"], r["C"], r["lda"], r["ldb"], r["ldc"], r["M"], r["N"], r["K"] = dY.data_ptr(), X.data_ptr(), C.data_ptr(), dY.shape[1], X.shape[1], X.shape[1], M, dY.shape[1], X.shape[1]
        nwg = int(os.environ.get("NWG", 256))
        Ms, Ns, Ks = (np.ascontiguousarray(rec[f], dtype=np.int32) for f in ("M", "N", "K"))
        cap = int(sum((n // 256) * (k // 256) for n, k in zip(Ns, Ks))) + 2 * nwg + 1
        items, first = np.zeros((cap, 8), dtype=np.int32), np.zeros(nwg + 1, dtype=np.int32)
        n = _lib.load().climb_tn_grouped_plan(len(flat), Ms.ctypes.data, Ns.ctypes.data, Ks.ctypes.data, nwg, items.ctypes.data, cap, first.ctypes.data)
        d = [torch.from_numpy(rec.view(np.uint8).copy()).to(dev), torch.from_numpy(items[:n].copy()).to(dev), torch.from_numpy(first).to(dev)]
        t = timeit(lambda: _lib.call("climb_gemm_bf16_tn_grouped", d[0], d[1], d[2], nwg, 0, st()), iters=10)
        f = sum(2.0 * M * N * K for N, K in shapes) * layers
        print(f"TN grouped, {layers:2d} layers per launch ({n} items, {int((items[:n, 5] == 1).sum())} partial): {t*1e6/layers:8.1f} us/layer  {f/t/1e12:7.1f} TF")
