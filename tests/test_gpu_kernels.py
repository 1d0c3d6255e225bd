"""Kernel-level parity on the MI355X: every C-ABI launcher against a plain PyTorch fp32 reference of the same op
(run on the CPU in float64 where it matters).  Tolerances are written next to each check."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _st():
    return torch.cuda.current_stream().cuda_stream


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


@pytest.mark.parametrize("M,N,K", [(64, 64, 16), (370, 768, 768), (130, 2304, 768), (64, 3129, 1536), (7, 2, 1536), (300, 96, 52)])
def test_gemm_f32_forward_layout(M, N, K):
    from climb_amd import _lib
    dev = _dev()
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g)       # asymmetric operands: a transposed write would be caught
    bias = torch.randn(N, generator=g)
    ref = A.double() @ W.double().t() + bias.double()
    Ad, Wd, bd = A.to(dev), W.to(dev), bias.to(dev)
    C = torch.empty(M, N, device=dev)
    _lib.call("climb_gemm_f32", Ad, K, 1, Wd, K, 1, C, N, M, N, K, bd, 0, None, 0, None, 0, 0.0, None, 0, 0, _st())
    assert _rel(C, ref) < 2e-6


def test_gemm_f32_splitk_skinny():
    """task-head shapes (M = batch): split-K with atomic partial sums, beta = 0 (zeroed by the launcher) and beta = 1"""
    from climb_amd import _lib
    dev = _dev()
    g = torch.Generator().manual_seed(9)
    M, N, K = 64, 1536, 3129
    A = torch.randn(M, 3132, generator=g)[:, :K]
    W = torch.randn(K, N, generator=g) * 0.05         # B(n, k) = W[k*N + n]
    bias = torch.randn(N, generator=g)
    Ad, Wd = torch.zeros(M, 3132).copy_(torch.nn.functional.pad(A, (0, 3))).to(dev), W.to(dev)
    C = torch.full((M, N), 7.0, device=dev)
    _lib.call("climb_gemm_f32", Ad, 3132, 1, Wd, 1, N, C, N, M, N, K, bias.to(dev), 0, None, 0, None, 0, 0.0, None, 0, 1, _st())
    ref = A.double() @ W.double() + bias.double()
    assert _rel(C, ref) < 2e-6
    C2 = torch.ones(M, N, device=dev)
    _lib.call("climb_gemm_f32", Ad, 3132, 1, Wd, 1, N, C2, N, M, N, K, None, 0, None, 0, None, 0, 1.0, None, 0, 1, _st())
    assert _rel(C2, 1 + A.double() @ W.double()) < 2e-6


def test_gemm_f32_strided_variants_and_epilogues():
    from climb_amd import _lib
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    M, N, K = 200, 192, 136
    dY = torch.randn(M, N, generator=g)
    W = torch.randn(N, K, generator=g)
    X = torch.randn(M, K, generator=g)
    U = torch.randn(M, K, generator=g)
    dYd, Wd, Xd, Ud = dY.to(dev), W.to(dev), X.to(dev), U.to(dev)
    # input grad with GELU' epilogue: dX = (dY W) * gelu'(U)
    dX = torch.empty(M, K, device=dev)
    _lib.call("climb_gemm_f32", dYd, N, 1, Wd, 1, K, dX, K, M, K, N, None, 3, Ud, K, None, 0, 0.0, None, 0, 0, _st())
    Ur = U.double().requires_grad_(True)
    gelu(Ur).backward(dY.double() @ W.double())
    assert _rel(dX, Ur.grad) < 5e-6
    # weight grad, accumulating: dW += dY^T X
    dW0 = torch.randn(N, K, generator=g)
    dW = dW0.to(dev).clone()
    _lib.call("climb_gemm_f32", dYd, 1, N, Xd, 1, K, dW, K, N, K, M, None, 0, None, 0, None, 0, 1.0, None, 0, 0, _st())
    assert _rel(dW, dW0.double() + dY.double().t() @ X.double()) < 2e-6
    # forward with GELU (pre-activation saved), residual and tanh epilogues
    b = torch.randn(N, generator=g).to(dev)
    Y = torch.empty(M, N, device=dev)
    pre = torch.empty(M, N, device=dev)
    _lib.call("climb_gemm_f32", Xd, K, 1, Wd, K, 1, Y, N, M, N, K, b, 1, None, 0, pre, N, 0.0, None, 0, 0, _st())
    ref_pre = X.double() @ W.double().t() + b.cpu().double()
    assert _rel(pre, ref_pre) < 2e-6 and _rel(Y, gelu(ref_pre)) < 5e-6
    R = torch.randn(M, N, generator=g).to(dev)
    _lib.call("climb_gemm_f32", Xd, K, 1, Wd, K, 1, Y, N, M, N, K, b, 2, R, N, None, 0, 0.0, None, 0, 0, _st())
    assert _rel(Y, ref_pre + R.cpu().double()) < 2e-6
    Xs = (X * 0.05).to(dev)                     # keep tanh out of saturation so the check is meaningful
    _lib.call("climb_gemm_f32", Xs, K, 1, Wd, K, 1, Y, N, M, N, K, b, 4, None, 0, None, 0, 0.0, None, 0, 0, _st())
    assert _rel(Y, torch.tanh((X * 0.05).double() @ W.double().t() + b.cpu().double())) < 5e-6


@pytest.mark.parametrize("C,dt", [(768, "f32"), (1536, "f32"), (768, "bf16")])
def test_layernorm_fwd_bwd(C, dt):
    from climb_amd import _lib
    dev = _dev()
    M = 77
    g = torch.Generator().manual_seed(C)
    x = torch.randn(M, C, generator=g) * 2 + 0.5
    gamma = 1 + 0.1 * torch.randn(C, generator=g)
    beta = 0.1 * torch.randn(C, generator=g)
    dy = torch.randn(M, C, generator=g)
    dres = torch.randn(M, C, generator=g)
    code = 0 if dt == "f32" else 1
    tdt = torch.float32 if dt == "f32" else _h16()
    tol = 2e-6 if dt == "f32" else 6e-3
    xd = x.to(dev)
    y = torch.empty(M, C, device=dev, dtype=tdt)
    mean = torch.empty(M, device=dev)
    rstd = torch.empty(M, device=dev)
    _lib.call("climb_layernorm_fwd", xd, C, gamma.to(dev), beta.to(dev), 1e-12, y, C, code, mean, rstd, M, C, _st())
    xr = x.double().requires_grad_(True)
    gr = gamma.double().requires_grad_(True)
    br = beta.double().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (C,), gr, br, 1e-12)
    assert _rel(y.float(), yr.detach()) < tol
    dy_used = dy.to(dev).to(tdt)
    yr.backward(dy_used.float().cpu().double())
    rows = _lib.query("climb_layernorm_bwd_rows_per_block")
    nb = (M + rows - 1) // rows
    part = torch.empty(nb * 3 * C, device=dev)
    dxo = torch.empty(M, C, device=dev)
    dcast = torch.empty(M, C, device=dev, dtype=tdt)
    _lib.call("climb_layernorm_bwd", dy_used, C, code, xd, C, mean, rstd, gamma.to(dev), dres.to(dev), C, dxo, C, dcast, C, part, M, C, _st())
    assert _rel(dxo, dres.double() + xr.grad) < 5e-6
    assert _rel(dcast.float(), dres.double() + xr.grad) < tol
    out = torch.zeros(3 * C, device=dev)
    _lib.call("climb_colreduce", part, 3 * C, nb, out, 3 * C, 0.0, _st())
    assert _rel(out[:C], gr.grad) < 5e-6
    assert _rel(out[C:2 * C], br.grad) < 5e-6
    assert _rel(out[2 * C:], (dres.double() + xr.grad).sum(0)) < 5e-6
    o0, o2 = torch.ones(C, device=dev), torch.ones(C, device=dev)           # fused triple, accumulating, middle output skipped
    _lib.call("climb_colreduce3", part, 3 * C, nb, o0, None, o2, C, 1.0, _st())
    assert _rel(o0, 1 + gr.grad) < 5e-6 and _rel(o2, 1 + (dres.double() + xr.grad).sum(0)) < 5e-6


def _attn_ref(qkv, bias, heads):
    B, S, H3 = qkv.shape
    H = H3 // 3
    d = H // heads
    q, k, v = qkv.split(H, dim=-1)
    q = q.view(B, S, heads, d).transpose(1, 2)
    k = k.view(B, S, heads, d).transpose(1, 2)
    v = v.view(B, S, heads, d).transpose(1, 2)
    s = q @ k.transpose(-1, -2) / math.sqrt(d) + bias[:, None, None, :]
    p = torch.softmax(s, dim=-1)
    return (p @ v).transpose(1, 2).reshape(B, S, H)


@pytest.mark.parametrize("M,N,K", [(2048, 768, 768), (2112, 264, 200), (4096, 2304, 768), (12288, 768, 3072), (12288, 3072, 768), (12288, 768, 768)])
def test_gemm_bf16_tn_persistent_tiles(M, N, K):
    """The persistent 256-row-tile weight-gradient kernel (option 10 = 1, the default for M >= 2048): against float64 of the same
    bf16-rounded operands where the host can afford it, and against the 128 x 128 kernel (itself pinned to float64 above) at the
    benchmark's 12288 tokens; accumulating into a non-zero C, fused bias gradient, ragged last tiles (264 x 200), uneven splits."""
    from climb_amd import _lib
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(M + 3 * N + K)
    dY = _bf(torch.randn(M, N, device=dev, generator=g))
    X = _bf(torch.randn(M, K, device=dev, generator=g))
    C0 = torch.randn(N, K, device=dev, generator=g)
    db0 = torch.randn(N, device=dev, generator=g)
    from climb_amd.engine import tn_workspace
    out = {}
    try:
        for mode in (2, 1, 0):                  # 2: persistent kernel, split partial sums through the registered scratch + reduce launch;
            _lib.call("climb_set_option", 10, min(mode, 1))      # 1: the same kernel with fp32 atomics (no scratch); 0: the 128 x 128 kernel
            if mode == 2:
                tn_workspace(dev)
            else:
                _lib.call("climb_set_tn_workspace", None, 0)
            C, db = C0.clone(), db0.clone()
            _lib.call("climb_gemm_bf16_tn", dY, N, X, K, C, K, M, N, K, db, _st())
            out[mode] = (C, db)
            for _ in range(2):                      # repeat: the counted-vmcnt schedule must give the same answer every time (to atomics' order)
                C2 = C0.clone()
                _lib.call("climb_gemm_bf16_tn", dY, N, X, K, C2, K, M, N, K, None, _st())
                assert _rel(C2, C) < 1e-5
    finally:
        _lib.call("climb_set_option", 10, 1)
        tn_workspace(dev)
    for mode in (2, 1):
        assert _rel(out[mode][0], out[0][0]) < 1e-5 and _rel(out[mode][1], out[0][1]) < 1e-5, mode
    if M * N * K <= 4096 * 2304 * 768:
        ref = C0.double().cpu() + dY.double().cpu().t() @ X.double().cpu()
        for mode in (2, 1):
            assert _rel(out[mode][0], ref) < 1e-5
            assert _rel(out[mode][1], db0.double().cpu() + dY.double().cpu().sum(0)) < 1e-5


def test_colreduce_batched_matches_single_reductions():
    """r03: the {dgamma, dbeta, bias} reductions of a group of LayerNorm backwards as one launch == one climb_colreduce3 per LayerNorm
    (bit for bit: same summation tree), accumulating into existing values, NULL outputs skipped, segments of different sizes."""
    from climb_amd import _lib
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(9)
    segs = [(768, 768, (True, True, True)), (768, 768, (True, True, False)), (96, 1536, (True, False, True)), (5, 40, (False, True, True))]
    parts = [torch.randn(nblk * 3 * nc, device=dev, generator=g) for nblk, nc, _ in segs]
    outs0 = [[torch.randn(nc, device=dev, generator=g) if w else None for w in want] for _, nc, want in segs]
    ref = [[o.clone() if o is not None else None for o in oo] for oo in outs0]
    for (nblk, nc, _), part, oo in zip(segs, parts, ref):
        _lib.call("climb_colreduce3", part, 3 * nc, nblk, oo[0], oo[1], oo[2], nc, 1.0, _st())
    got = [[o.clone() if o is not None else None for o in oo] for oo in outs0]
    rec = np.zeros(len(segs), dtype=[("part", "<u8"), ("stride", "<i8"), ("out", "<u8", (3,)), ("nblk", "<i4"), ("ncols", "<i4")])
    for r, (nblk, nc, _), part, oo in zip(rec, segs, parts, got):
        r["part"], r["stride"], r["nblk"], r["ncols"] = part.data_ptr(), 3 * nc, nblk, nc
        r["out"] = [o.data_ptr() if o is not None else 0 for o in oo]
    d = torch.from_numpy(rec.view(np.uint8).copy()).to(dev)
    _lib.call("climb_colreduce_batched", d, len(segs), 1536, _st())
    torch.cuda.synchronize()
    for a, b in zip(got, ref):
        for x, y in zip(a, b):
            assert (x is None) == (y is None)
            if x is not None:
                assert torch.equal(x, y)


@pytest.mark.parametrize("ragged", [0, 1])
@pytest.mark.parametrize("nwg", [256, 8, 24])
def test_gemm_bf16_tn_grouped_launch(nwg, ragged):
    """The grouped weight-gradient launch (r03: several dW GEMMs in one persistent kernel, whole tiles reduce over all tokens and
    read-modify-write C, the stream-K tail meets through atomics) against float64 of the same 16-bit operands: problems of different
    token counts and shapes, accumulation into non-zero C, fused bias gradients on some, nwg = 256 (everything is tail), 8 and 24 (whole
    rounds + tail).  Repeated launches must agree (to the atomics' order)."""
    from climb_amd import _lib
    dev = _dev()
    # C / dbias carry guard rows and columns behind the ragged problems' edges? No: their buffers are exactly [N, K] / [N] -- a write outside
    # would corrupt the neighbouring allocation and show up in ITS comparison below
    shapes = [(2048, 768, 768, True), (2048, 256, 512, False), (1024, 512, 256, True), (3072, 256, 256, False), (1152, 768, 256, True)]
    if ragged:          # the adapters' shapes and an odd one: surplus tile columns are computed on clamped addresses and never stored
        shapes += [(2048, 768, 48, True), (2048, 48, 768, True), (1024, 264, 200, False)]
    g = torch.Generator(device=dev).manual_seed(5 + nwg)
    ops = []
    for M, N, K, bias in shapes:
        ops.append((_bf(torch.randn(M, N, device=dev, generator=g)), _bf(torch.randn(M, K, device=dev, generator=g)),
                    torch.randn(N, K, device=dev, generator=g), torch.randn(N, device=dev, generator=g) if bias else None))
    rec = np.zeros(len(shapes), dtype=[("A", "<u8"), ("B", "<u8"), ("C", "<u8"), ("dbias", "<u8"), ("lda", "<i8"), ("ldb", "<i8"), ("ldc", "<i8"),
                                       ("M", "<i4"), ("N", "<i4"), ("K", "<i4"), ("reserved", "<i4")])

    def run():
        Cs = [c.clone() for _, _, c, _ in ops]
        dbs = [b.clone() if b is not None else None for *_, b in ops]
        for r, (M, N, K, _), (dY, X, _, _), C, db in zip(rec, shapes, ops, Cs, dbs):
            r["A"], r["B"], r["C"], r["dbias"] = dY.data_ptr(), X.data_ptr(), C.data_ptr(), (db.data_ptr() if db is not None else 0)
            r["lda"], r["ldb"], r["ldc"], r["M"], r["N"], r["K"] = N, K, K, M, N, K
        Ms, Ns, Ks = (np.ascontiguousarray(rec[f], dtype=np.int32) for f in ("M", "N", "K"))
        cap = int(sum(((n + 255) // 256) * ((k + 255) // 256) for n, k in zip(Ns, Ks))) + 2 * nwg + 1
        items, first = np.zeros((cap, 8), dtype=np.int32), np.zeros(nwg + 1, dtype=np.int32)
        n = _lib.load().climb_tn_grouped_plan(len(shapes), Ms.ctypes.data, Ns.ctypes.data, Ks.ctypes.data, nwg, items.ctypes.data, cap, first.ctypes.data)
        assert n > 0
        d_rec = torch.from_numpy(rec.view(np.uint8).copy()).to(dev)
        d_items, d_first = torch.from_numpy(items[:n].copy()).to(dev), torch.from_numpy(first).to(dev)
        _lib.call("climb_gemm_bf16_tn_grouped", d_rec, d_items, d_first, nwg, ragged, _st())
        torch.cuda.synchronize()
        return Cs, dbs, items[:n]
    Cs, dbs, items = run()
    if nwg != 256:
        assert (items[:, 5] == 0).any() and (items[:, 5] == 1).any()        # both epilogues ran
    for (dY, X, C0, b0), C, db in zip(ops, Cs, dbs):
        ref = C0.double().cpu() + dY.double().cpu().t() @ X.double().cpu()
        assert _rel(C, ref) < 1e-5
        if b0 is not None:
            assert _rel(db, b0.double().cpu() + dY.double().cpu().sum(0)) < 1e-5
    Cs2, dbs2, _ = run()
    for a, b in zip(Cs, Cs2):
        assert _rel(a, b) < 1e-5


@pytest.mark.parametrize("dirty", [False, True])
def test_gemm_bf16_tn_grouped_with_adamw_in_the_epilogue_equals_launch_then_flat_adamw(dirty):
    """r04: climb_gemm_bf16_tn_grouped_adamw applies AdamW in the epilogue of every WHOLE tile of a problem marked `fused` (p, m, v, the 16-bit shadow
    and the transposed shadow written from the tile sum; the gradient never stored), everything else as climb_gemm_bf16_tn_grouped.  Against
    { climb_gemm_bf16_tn_grouped ; climb_adamw ; climb_transpose_bf16 } on the same operands: the update function is shared and the tile sums are the
    same sums, so every output of a fused problem must be BIT-IDENTICAL.  Problems: 2 x [768,3072] + [2304,768] + 2 x [768,768] + [256,512] = 119
    tiles on 64 workgroups: one whole round and a stream-K tail; the problems the tail touches are NOT fused (their gradients land in C as before).
    dirty: the gradient buffer already holds a term (the EWC penalty's, REF/cl_algorithms/ewc.py:75-87) that the fused update must add."""
    from climb_amd import _lib
    if _lib.torch_h16() != torch.bfloat16:
        pytest.skip("bf16 build only")
    dev = _dev()
    M, nwg = 1024, 64
    shapes = [(768, 3072), (768, 3072), (2304, 768), (768, 768), (768, 768), (256, 512)]
    g = torch.Generator(device=dev).manual_seed(11)
    total = sum(n * k for n, k in shapes)
    offs = np.cumsum([0] + [n * k for n, k in shapes])
    mk = lambda scale=1.0: torch.randn(total, device=dev, generator=g) * scale
    P0, M0, V0 = mk(0.05), mk(0.01), (mk(0.01)).abs()
    G0 = mk(0.02) if dirty else torch.zeros(total, device=dev)
    dYs = [torch.randn(M, n, device=dev, generator=g).to(torch.bfloat16) for n, k in shapes]
    Xs = [torch.randn(M, k, device=dev, generator=g).to(torch.bfloat16) for n, k in shapes]
    Ms, Ns, Ks = (np.array(v, dtype=np.int32) for v in ([M] * len(shapes), [n for n, _ in shapes], [k for _, k in shapes]))
    cap = int(sum((n // 256) * (k // 256) for n, k in shapes)) + 2 * nwg + 1
    items, first = np.zeros((cap, 8), dtype=np.int32), np.zeros(nwg + 1, dtype=np.int32)
    n_items = _lib.load().climb_tn_grouped_plan(len(shapes), Ms.ctypes.data, Ns.ctypes.data, Ks.ctypes.data, nwg, items.ctypes.data, cap, first.ctypes.data)
    assert n_items > 0
    partial_probs = set(int(x) for x in items[:n_items][items[:n_items, 5] == 1, 0])
    assert partial_probs and len(partial_probs) < len(shapes)
    fused = [i not in partial_probs for i in range(len(shapes))]
    adam = np.array([1e-3, 1e-2, 0.9, 0.98, 1e-8, 1 - 0.9 ** 3, 1 - 0.98 ** 3, 1.0], dtype=np.float32)

    def run(use_fused):
        P, Mm, Vv, G = P0.clone(), M0.clone(), V0.clone(), G0.clone()
        S = P.to(torch.bfloat16)
        ST = torch.zeros(total, device=dev, dtype=torch.bfloat16)
        rec = np.zeros(len(shapes), dtype=[("A", "<u8"), ("B", "<u8"), ("C", "<u8"), ("dbias", "<u8"), ("lda", "<i8"), ("ldb", "<i8"), ("ldc", "<i8"),
                                           ("M", "<i4"), ("N", "<i4"), ("K", "<i4"), ("reserved", "<i4")])
        opt = np.zeros(len(shapes), dtype=[("p", "<u8"), ("m", "<u8"), ("v", "<u8"), ("s", "<u8"), ("st", "<u8"), ("ldt", "<i8"), ("fused", "<i4"), ("pad", "<i4")])
        for i, (n, k) in enumerate(shapes):
            o = int(offs[i])
            rec[i]["A"], rec[i]["B"], rec[i]["C"] = dYs[i].data_ptr(), Xs[i].data_ptr(), G.data_ptr() + 4 * o
            rec[i]["lda"], rec[i]["ldb"], rec[i]["ldc"], rec[i]["M"], rec[i]["N"], rec[i]["K"] = n, k, k, M, n, k
            opt[i]["p"], opt[i]["m"], opt[i]["v"] = P.data_ptr() + 4 * o, Mm.data_ptr() + 4 * o, Vv.data_ptr() + 4 * o
            opt[i]["s"], opt[i]["st"], opt[i]["ldt"], opt[i]["fused"] = S.data_ptr() + 2 * o, ST.data_ptr() + 2 * o, n, int(fused[i])
        d_rec = torch.from_numpy(rec.view(np.uint8).copy()).to(dev)
        d_opt = torch.from_numpy(opt.view(np.uint8).copy()).to(dev)
        d_items, d_first = torch.from_numpy(items[:n_items].copy()).to(dev), torch.from_numpy(first).to(dev)
        starts = torch.from_numpy(np.asarray(offs, dtype=np.int64)).to(dev)
        if use_fused:
            _lib.call("climb_gemm_bf16_tn_grouped_adamw", d_rec, d_items, d_first, nwg, 0, d_opt, adam.ctypes.data, 1 if dirty else 0, _st())
            groups = torch.tensor([-1 if f else 0 for f in fused], dtype=torch.int8, device=dev)
        else:
            _lib.call("climb_gemm_bf16_tn_grouped", d_rec, d_items, d_first, nwg, 0, _st())
            groups = torch.zeros(len(shapes), dtype=torch.int8, device=dev)
        table = adam.copy().reshape(1, 8)
        table[0, 7] = 0.0
        _lib.call("climb_adamw", P, G, Mm, Vv, S, total, starts, groups, len(shapes), table.ctypes.data, 1, 1.0, _st())
        for i, (n, k) in enumerate(shapes):
            if not (use_fused and fused[i]):
                o = int(offs[i])
                _lib.call("climb_transpose_bf16", S[o:o + n * k], ST[o:o + n * k], n, k, _st())
        torch.cuda.synchronize()
        return P, Mm, Vv, S, ST, G
    ref, got = run(False), run(True)
    for name, a, b in zip(("p", "m", "v", "shadow", "transposed shadow"), ref, got):
        for i, (n, k) in enumerate(shapes):
            if fused[i]:
                o = int(offs[i])
                bad = int((a[o:o + n * k] != b[o:o + n * k]).sum())
                assert bad == 0, f"{name} of fused problem {i} {shapes[i]}: {bad} elements differ, max |d| {float((a[o:o + n * k].float() - b[o:o + n * k].float()).abs().max()):.3e} (max |ref| {float(a[o:o + n * k].float().abs().max()):.3e})"
    # the update is really AdamW of the true gradient (float64 spot check of one fused problem), and its gradient range was never written
    i = fused.index(True)
    n, k = shapes[i]
    o = int(offs[i])
    gt = dYs[i].double().t() @ Xs[i].double() + G0[o:o + n * k].view(n, k).double()
    lr, wd, b1, b2, eps, bc1, bc2, _ = [float(x) for x in adam]
    mm = b1 * M0[o:o + n * k].view(n, k).double() + (1 - b1) * gt
    vv = b2 * V0[o:o + n * k].view(n, k).double() + (1 - b2) * gt * gt
    pp = P0[o:o + n * k].view(n, k).double() * (1 - lr * wd) - (lr / bc1) * mm / (vv.sqrt() / math.sqrt(bc2) + eps)
    assert _rel(got[0][o:o + n * k].view(n, k), pp) < 1e-5 and _rel(got[1][o:o + n * k].view(n, k), mm) < 1e-4
    assert torch.equal(got[5][o:o + n * k], G0[o:o + n * k])


@pytest.mark.parametrize("S_pad,valid", [(64, 50), (192, 185), (224, 200), (288, 281)])
def test_attention_f32_fwd_bwd(S_pad, valid):
    from climb_amd import _lib
    dev = _dev()
    B, heads, d = 2, 3, 64
    H = heads * d
    g = torch.Generator().manual_seed(S_pad)
    qkv = torch.randn(B, S_pad, 3 * H, generator=g)
    qkv[..., :2 * H] *= 1.5                       # non-trivial softmax
    bias = torch.zeros(B, S_pad)
    bias[:, valid:] = -3.0e38
    bias[1, 3:7] = -3.0e38                        # masked text tokens in the middle
    dctx = torch.randn(B, S_pad, H, generator=g)
    dctx[:, valid:] = 0
    qr = qkv.double().requires_grad_(True)
    ref = _attn_ref(qr, bias.double().clamp(min=-1e300), heads)
    ref.backward(dctx.double())
    qd, bd, dd = qkv.to(dev).view(B * S_pad, 3 * H), bias.to(dev), dctx.to(dev).view(B * S_pad, H)
    ctx = torch.empty(B * S_pad, H, device=dev)
    lse = torch.empty(B, heads, S_pad, device=dev)
    _lib.call("climb_attn_fwd_f32", qd, bd, ctx, lse, B, S_pad, heads, d, _st())
    assert _rel(ctx.view(B, S_pad, H)[:, :valid], ref.detach()[:, :valid]) < 5e-6
    delta = torch.empty(B, heads, S_pad, device=dev)
    dqkv = torch.full((B * S_pad, 3 * H), float("nan"), device=dev)
    _lib.call("climb_attn_delta", dd, ctx, 0, delta, B, S_pad, heads, _st())
    _lib.call("climb_attn_bwd_f32", qd, bd, dd, lse, delta, dqkv, B, S_pad, heads, d, _st())
    assert not torch.isnan(dqkv).any()
    assert _rel(dqkv.view(B, S_pad, 3 * H), qr.grad) < 1e-5


def test_losses():
    from climb_amd import _lib
    dev = _dev()
    g = torch.Generator().manual_seed(0)
    B, N = 5, 3129
    x = torch.randn(B, N, generator=g) * 3
    t = (torch.rand(B, N, generator=g) > 0.999).float() * 0.6
    xr = x.double().requires_grad_(True)
    ref = torch.nn.functional.binary_cross_entropy_with_logits(xr, t.double(), reduction="mean") * N
    ref.backward()
    loss = torch.empty((), device=dev)
    dl = torch.empty(B, N, device=dev)
    ws = torch.empty(_lib.query("climb_bce_workspace_floats"), device=dev)
    _lib.call("climb_bce_logits", x.to(dev), N, t.to(dev), N, dl, N, loss, ws, B, N, 1.0, _st())
    assert abs(loss.item() - ref.item()) < 2e-6 * abs(ref.item())
    assert _rel(dl, xr.grad) < 2e-6
    for n in (2, 3, 4):
        x = torch.randn(B, n, generator=g)
        lab = torch.randint(0, n, (B,), generator=g)
        xr = x.double().requires_grad_(True)
        ref = torch.nn.functional.cross_entropy(xr, lab)
        ref.backward()
        dl = torch.empty(B, n, device=dev)
        _lib.call("climb_cross_entropy", x.to(dev), n, lab.to(dev), dl, n, loss, B, n, 1.0, _st())
        assert abs(loss.item() - ref.item()) < 2e-6 and _rel(dl, xr.grad) < 2e-6


@pytest.mark.parametrize("M,N,K,kcontig,epi", [(64, 768, 768, True, 1), (64, 1536, 768, True, 0), (64, 3129, 1536, True, 0), (64, 1536, 3129, False, 3),
                                                (64, 768, 1536, False, 2), (32, 1536, 1536, True, 0), (5, 3, 70, True, 0), (130, 40, 33, False, 1)])
def test_skinny_f32_products_of_the_heads(M, N, K, kcontig, epi):
    """r04 csrc/heads.hip: C = epi(A B^T + bias) with every row of a 16-column strip in one workgroup (K split over 8 waves, summed in LDS in a
    fixed order), against float64: the pooler / head shapes at the benchmark's batch (K = 3129 is not a multiple of the 16-deep k-block), both B
    layouts (weights [N, K] and transposed use [K, N]), every epilogue, ragged M / N / K, M > 64 (two row tiles); the column sums of C and of A
    (bias gradients, M <= 64) are added to what the target held; two launches give identical bits."""
    from climb_amd import _lib
    dev = _dev()
    g = torch.Generator().manual_seed(M * 7 + N + K)
    lda = (K + 3) // 4 * 4
    A = torch.zeros(M, lda)
    A[:, :K] = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    aux = torch.randn(M, N, generator=g)
    ref = A[:, :K].double() @ W.double().t() + bias.double()
    if epi == 1:
        ref = torch.tanh(ref)
    elif epi == 2:
        ref = ref * (1 - aux.double() ** 2)
    elif epi == 3:
        x = aux.double()
        ref = ref * (0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi))
    Ad, auxd, bd = A.to(dev), aux.to(dev), bias.to(dev)
    if kcontig:
        Kp = (K + 3) // 4 * 4
        Wd = torch.zeros(N, Kp, device=dev)
        Wd[:, :K] = W.to(dev)
        sbn, sbk = Kp, 1
    else:
        Wd = W.t().contiguous().to(dev)          # [K, N]: B(n, k) = Wd[k, n]
        sbn, sbk = 1, N
    want_sums = M <= 64
    cs0, ac0 = torch.randn(N, generator=g), torch.randn(K, generator=g)
    outs = []
    for _ in range(2):
        C = torch.full((M, N + 3), float("nan"), device=dev)
        cs, ac = cs0.to(dev), ac0.to(dev)
        _lib.call("climb_skinny_f32", Ad, lda, Wd, sbn, sbk, C, N + 3, M, N, K, bd, epi, auxd if epi >= 2 else None, N, cs if want_sums else None, 1.0,
                  ac if want_sums else None, 1.0, _st())
        outs.append((C.cpu(), cs.cpu(), ac.cpu()))
    C, cs, ac = outs[0]
    assert torch.isnan(C[:, N:]).all()                       # nothing outside the N columns
    assert _rel(C[:, :N], ref) < 3e-6
    if want_sums:
        assert _rel(cs, cs0.double() + ref.sum(0)) < 3e-6
        assert _rel(ac, ac0.double() + A[:, :K].double().sum(0)) < 3e-6
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a[:, :N] if a.dim() == 2 else a, b[:, :N] if b.dim() == 2 else b)
    if not want_sums:
        with pytest.raises(RuntimeError):
            _lib.call("climb_skinny_f32", Ad, lda, Wd, sbn, sbk, C.to(dev), N + 3, M, N, K, bd, 0, None, 0, cs0.to(dev), 1.0, None, 1.0, _st())


@pytest.mark.parametrize("M,N,K", [(64, 3129, 1536), (64, 1536, 768), (32, 768, 768), (5, 70, 33), (100, 130, 64)])
def test_rank_update_f32_weight_gradients_of_the_heads(M, N, K):
    """C[n, k] += sum_m dY[m, n] X[m, k] (csrc/heads.hip) against float64, added onto what C held, ragged tiles, more than 64 rows, strided operands;
    nothing is written outside the N x K block; two launches from the same start give identical bits."""
    from climb_amd import _lib
    dev = _dev()
    g = torch.Generator().manual_seed(M + N + K)
    lddy, ldx, ldc = N + 3, K + 1, K + 2
    dY = torch.randn(M, lddy, generator=g)
    X = torch.randn(M, ldx, generator=g)
    C0 = torch.randn(N, ldc, generator=g)
    ref = C0[:, :K].double() + dY[:, :N].double().t() @ X[:, :K].double()
    outs = []
    for _ in range(2):
        C = C0.to(dev)
        _lib.call("climb_rank_update_f32", dY.to(dev), lddy, X.to(dev), ldx, C, ldc, M, N, K, _st())
        outs.append(C.cpu())
    assert _rel(outs[0][:, :K], ref) < 3e-6
    assert torch.equal(outs[0][:, K:], C0[:, K:]) and torch.equal(outs[0], outs[1])


def test_layernorm_gelu_forward_of_the_head():
    """zn = LayerNorm(z) and gz = gelu(zn) in one pass (REF/modeling/vilt.py:191-193) = climb_layernorm_fwd followed by the gelu pass, bit for bit"""
    from climb_amd import _lib
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    for M, C in ((64, 1536), (7, 768), (33, 1536)):
        z = (torch.randn(M, C, generator=g) * 2 + 0.3).to(dev)
        gam, bet = torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
        zn, gz, mu, rs = (torch.empty(M, C, device=dev), torch.empty(M, C, device=dev), torch.empty(M, device=dev), torch.empty(M, device=dev))
        _lib.call("climb_layernorm_gelu_fwd", z, C, gam, bet, 1e-5, zn, gz, C, mu, rs, M, C, _st())
        zn2, gz2, mu2, rs2 = (torch.empty(M, C, device=dev), torch.empty(M, C, device=dev), torch.empty(M, device=dev), torch.empty(M, device=dev))
        _lib.call("climb_layernorm_fwd", z, C, gam, bet, 1e-5, zn2, C, 0, mu2, rs2, M, C, _st())
        _lib.call("climb_elementwise", 0, zn2, None, gz2, M * C, 1.0, _st())
        assert torch.equal(zn, zn2) and torch.equal(gz, gz2) and torch.equal(mu, mu2) and torch.equal(rs, rs2)
        ref = torch.nn.functional.gelu(torch.nn.functional.layer_norm(z.double().cpu(), (C,), gam.double().cpu(), bet.double().cpu(), 1e-5))
        assert _rel(gz, ref) < 3e-6


@pytest.mark.parametrize("zero,from16", [(0, False), (1, False), (0, True)])
def test_adamw_spans_equals_the_flat_pass(zero, from16):
    """r04: climb_adamw_spans walks only the maximal runs of tensors that have a group (spans in 1024-element blocks, ragged ends, a span of
    several tensors with different groups) and must update them bit for bit like climb_adamw; tensors outside the spans keep p, m, v AND g;
    with zero_grad = 1 the consumed gradients are cleared, nothing else is.  from16: one span takes its gradient from a 16-bit payload buffer
    times a scale (the data-parallel reducer's averaged payload) -- the same bits as casting that payload back with climb_uncast_bf16_scale and
    running the flat pass."""
    from climb_amd import _lib
    dev = _dev()
    g = torch.Generator().manual_seed(5 + zero)
    sizes = [64 * 3, 64 * 40, 64 * 17, 64 * 1, 64 * 100, 64 * 2, 64 * 33]
    groups = np.array([0, -1, 1, 0, -1, -1, 1], dtype=np.int8)
    starts = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(starts[-1])
    table = np.array([[1e-3, 1e-2, 0.9, 0.98, 1e-8, 1 - 0.9 ** 3, 1 - 0.98 ** 3, 0], [2e-3, 0.0, 0.9, 0.98, 1e-8, 1 - 0.9, 1 - 0.98, 0]], dtype=np.float32)
    p0, g0 = torch.randn(n, generator=g), torch.randn(n, generator=g) * 0.1
    m0, v0 = torch.rand(n, generator=g) * 0.01, torch.rand(n, generator=g) * 0.001
    d_starts, d_groups = torch.from_numpy(starts).to(dev), torch.from_numpy(groups).to(dev)
    spans, nb = [], 0
    for si in np.flatnonzero(groups >= 0):
        a, b = int(starts[si]), int(starts[si + 1])
        if spans and spans[-1][0] + spans[-1][1] == a:
            spans[-1][1] += b - a
        else:
            spans.append([a, b - a, 0, 0])
    assert len(spans) == 3 and spans[1][1] == 64 * 18          # tensors 2 and 3 (different groups) form one span
    stage, scale16 = None, 1.0
    gref = g0.clone()
    if from16:          # the middle span's gradient lives in the payload buffer: cut it in two (only the first part from there) to walk a source boundary
        a, ln = spans[1][0], spans[1][1]
        cut = a + 64 * 5
        spans[1:2] = [[a, cut - a, 0, 1], [cut, a + ln - cut, 0, 0]]
        stage = (torch.randn(n, generator=g) * 0.1).to(_h16()).to(dev)
        scale16 = 0.5
        tmp = torch.zeros(n, device=dev)
        _lib.call("climb_uncast_bf16_scale", stage[a:cut], tmp[a:cut], cut - a, scale16, _st())
        gref[a:cut] = tmp[a:cut].cpu()
    for sp in spans:
        sp[2] = nb
        nb += (sp[1] + 1023) // 1024
    ref = [t.to(dev).clone() for t in (p0, gref, m0, v0)]
    sh_ref = torch.zeros(n, device=dev, dtype=_h16())
    _lib.call("climb_adamw", ref[0], ref[1], ref[2], ref[3], sh_ref, n, d_starts, d_groups, len(sizes), table.ctypes.data, 2, 1.0, _st())
    out = [t.to(dev).clone() for t in (p0, g0, m0, v0)]
    sh = torch.zeros(n, device=dev, dtype=_h16())
    _lib.call("climb_adamw_spans", out[0], out[1], out[2], out[3], sh, torch.tensor(spans, dtype=torch.int64, device=dev).reshape(-1), len(spans), nb,
              d_starts, d_groups, len(sizes), table.ctypes.data, 2, 1.0, zero, stage, scale16, _st())
    for a, b in ((out[0], ref[0]), (out[2], ref[2]), (out[3], ref[3]), (sh, sh_ref)):
        assert torch.equal(a, b)
    gexp = g0.clone()
    if zero:
        for si in np.flatnonzero(groups >= 0):
            gexp[int(starts[si]):int(starts[si + 1])] = 0
    assert torch.equal(out[1].cpu(), gexp)
    with pytest.raises(RuntimeError):          # no spans: an argument error, not a launch
        _lib.call("climb_adamw_spans", out[0], out[1], out[2], out[3], None, None, 0, 0, d_starts, d_groups, len(sizes), table.ctypes.data, 2, 1.0, 0, None, 1.0, _st())


def test_adamw_ewc_fisher_flat_kernels():
    from climb_amd import _lib
    dev = _dev()
    g = torch.Generator().manual_seed(1)
    n = 64 * 40
    p = torch.randn(n, generator=g)
    gr = torch.randn(n, generator=g) * 0.1
    starts = np.array([0, 640, 1280, n], dtype=np.int64)
    groups = np.array([0, -1, 1], dtype=np.int8)
    table = np.array([[1e-3, 1e-2, 0.9, 0.98, 1e-8, 1 - 0.9 ** 3, 1 - 0.98 ** 3, 0], [2e-3, 0.0, 0.9, 0.98, 1e-8, 1 - 0.9, 1 - 0.98, 0]], dtype=np.float32)
    m0, v0 = torch.rand(n, generator=g) * 0.01, torch.rand(n, generator=g) * 0.001
    pd, gd, md, vd = p.to(dev), gr.to(dev), m0.to(dev), v0.to(dev)
    _lib.call("climb_adamw", pd, gd, md, vd, None, n, torch.from_numpy(starts).to(dev), torch.from_numpy(groups).to(dev), 3,
              table.ctypes.data, 2, 1.0, _st())
    ref = p.double().clone()
    for (lo, hi, row) in ((0, 640, 0), (1280, n, 1)):
        lr, wd, b1, b2, eps, bc1, bc2, _ = [float(x) for x in table[row]]
        pp, gg = ref[lo:hi], gr[lo:hi].double()
        mm = b1 * m0[lo:hi].double() + (1 - b1) * gg
        vv = b2 * v0[lo:hi].double() + (1 - b2) * gg * gg
        pp.mul_(1 - lr * wd)
        pp.sub_((lr / bc1) * mm / (vv.sqrt() / math.sqrt(bc2) + eps))
    assert _rel(pd, ref) < 2e-6
    assert torch.equal(pd[640:1280].cpu(), p[640:1280])          # skipped tensor untouched
    # EWC penalty + gradient, Fisher accumulate
    star, fis = torch.randn(n, generator=g), torch.rand(n, generator=g)
    grad = torch.randn(n, generator=g)
    gd2 = grad.to(dev)
    ws = torch.empty(_lib.query("climb_ewc_workspace_floats"), device=dev)
    out = torch.empty((), device=dev)
    _lib.call("climb_ewc_penalty", p.to(dev), star.to(dev), fis.to(dev), gd2, n, 100.0, 1.0, ws, out, _st())
    ref_l = 100.0 * (fis.double() * (p.double() - star.double()) ** 2).sum()
    assert abs(out.item() - ref_l.item()) < 2e-6 * ref_l.item()
    assert _rel(gd2, grad.double() + 200.0 * fis.double() * (p.double() - star.double())) < 2e-6
    f2 = fis.to(dev).clone()
    _lib.call("climb_fisher_accum", f2, gd2, n, _st())
    assert _rel(f2, fis.double() + gd2.cpu().double() ** 2) < 2e-6


# ------------------------------------------------------------------------------------------------ bf16 throughput path
def _h16():
    """torch dtype of the loaded library's 16-bit operand type: bf16, or IEEE half when the suite runs as CLIMB_AMD_H16=fp16 (the second
    build of the same sources; tests/test_gpu_fp16_build.py runs this file that way in a subprocess)."""
    from climb_amd import _lib
    return _lib.torch_h16()


def _bf(x):
    return x.to(_h16())


# the last two shapes select the 192x192 three-stage kernel (160..256 tiles), with a ragged last row tile
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (384, 768, 768), (200, 2304, 768), (390, 48, 768), (384, 768, 3072), (9184, 768, 768),
                                   (7712, 768, 1536)])
def test_gemm_bf16_nt(M, N, K):
    """bf16 operands, fp32 accumulate: compared with a float64 product of the SAME bf16-rounded operands, so the only
    error left is accumulation order + the bf16 rounding of the output (<= 2^-8 relative)."""
    from climb_amd import _lib
    dev = _dev()
    g = torch.Generator().manual_seed(M + N + K)
    A, W = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(N, K, generator=g) * 0.05)
    bias = torch.randn(N, generator=g)
    ref = A.double() @ W.double().t() + bias.double()
    Ad, Wd, bd = A.to(dev), W.to(dev), bias.to(dev)
    C32 = torch.empty(M, N, device=dev)
    _lib.call("climb_gemm_bf16_nt", Ad, K, Wd, K, C32, N, 0, M, N, K, bd, 0, None, 0, None, 0, None, 0, _st())
    assert _rel(C32, ref) < 1e-5
    C16 = torch.empty(M, N, device=dev, dtype=_h16())
    U = torch.empty(M, N, device=dev, dtype=_h16())
    _lib.call("climb_gemm_bf16_nt", Ad, K, Wd, K, C16, N, 1, M, N, K, bd, 1, None, 0, U, N, None, 0, _st())
    assert _rel(U.float(), ref) < 5e-3 and _rel(C16.float(), gelu(ref)) < 5e-3
    R = torch.randn(M, N, generator=g).to(dev)
    _lib.call("climb_gemm_bf16_nt", Ad, K, Wd, K, C32, N, 0, M, N, K, bd, 2, R, N, None, 0, None, 0, _st())
    assert _rel(C32, ref + R.cpu().double()) < 1e-5
    Uin = _bf(torch.randn(M, N, generator=g)).to(dev)
    _lib.call("climb_gemm_bf16_nt", Ad, K, Wd, K, C16, N, 1, M, N, K, None, 3, Uin, N, None, 0, None, 0, _st())
    ur = Uin.cpu().double().requires_grad_(True)
    gelu(ur).backward(A.double() @ W.double().t())
    assert _rel(C16.float(), ur.grad) < 5e-3
    _check_saved_derivative_pair(Ad, Wd, bd, Uin, ref, A.double() @ W.double().t(), M, N, K)


def _check_saved_derivative_pair(Ad, Wd, bd, Uin, ref, ref_nobias, M, N, K):
    """r05, epilogue codes 8 / 9: the forward saves gelu'(pre-activation) next to gelu(pre-activation), the backward multiplies by what was saved."""
    from climb_amd import _lib
    dev = Ad.device
    C16 = torch.empty(M, N, device=dev, dtype=_h16())
    D = torch.empty(M, N, device=dev, dtype=_h16())
    _lib.call("climb_gemm_bf16_nt", Ad, K, Wd, K, C16, N, 1, M, N, K, bd, 8, None, 0, D, N, None, 0, _st())
    pre = ref.clone().requires_grad_(True)
    gelu(pre).sum().backward()
    assert _rel(C16.float(), gelu(ref)) < 5e-3
    assert float((D.float().cpu().double() - pre.grad).abs().max()) < 5e-3          # gelu' is O(1): 16-bit rounding of the stored value + 1.1e-4 of the sigmoid form
    _lib.call("climb_gemm_bf16_nt", Ad, K, Wd, K, C16, N, 1, M, N, K, None, 9, Uin, N, None, 0, None, 0, _st())
    assert _rel(C16.float(), ref_nobias * Uin.cpu().double()) < 5e-3
    # no fp32 outputs for these codes, and the operands they need are checked
    C32 = torch.empty(M, N, device=dev)
    with pytest.raises(RuntimeError, match="climb_gemm_bf16_nt"):
        _lib.call("climb_gemm_bf16_nt", Ad, K, Wd, K, C32, N, 0, M, N, K, None, 8, None, 0, D, N, None, 0, _st())
    with pytest.raises(RuntimeError, match="climb_gemm_bf16_nt"):
        _lib.call("climb_gemm_bf16_nt", Ad, K, Wd, K, C16, N, 1, M, N, K, None, 9, None, 0, None, 0, None, 0, _st())


@pytest.mark.parametrize("force", [2, 3, 4])
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 328, 192), (1000, 2304, 768), (512, 3072, 768), (256, 768, 3072), (777, 512, 2304)])
def test_gemm_bf16_nt_256_tiles(M, N, K, force):
    """The persistent 256 x 256 (option 7 = 2) / 256 x 192 (= 3) / two-workgroup 128 x 192 (= 4) counted-vmcnt kernels forced on at shapes that walk its edges: the
    minimum of two k-tiles, an odd k-tile count, ragged last row / column tiles, every epilogue it implements, both output types --
    against float64 of the same bf16-rounded operands."""
    from climb_amd import _lib
    dev = _dev()
    g = torch.Generator().manual_seed(M + N + K)
    A, W = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(N, K, generator=g) * 0.05)
    bias = torch.randn(N, generator=g)
    ref = A.double() @ W.double().t() + bias.double()
    Ad, Wd, bd = A.to(dev), W.to(dev), bias.to(dev)
    _lib.call("climb_set_option", 7, force)
    try:
        C32 = torch.full((M + 1, N), 7.0, device=dev)          # one guard row: a tile that stores past M would be seen
        _lib.call("climb_gemm_bf16_nt", Ad, K, Wd, K, C32, N, 0, M, N, K, bd, 0, None, 0, None, 0, None, 0, _st())
        assert _rel(C32[:M], ref) < 1e-5 and bool((C32[M] == 7.0).all())
        C16 = torch.empty(M, N, device=dev, dtype=_h16())
        U = torch.empty(M, N, device=dev, dtype=_h16())
        _lib.call("climb_gemm_bf16_nt", Ad, K, Wd, K, C16, N, 1, M, N, K, bd, 1, None, 0, U, N, None, 0, _st())
        assert _rel(U.float(), ref) < 5e-3 and _rel(C16.float(), gelu(ref)) < 5e-3
        R = torch.randn(M, N, generator=g).to(dev)
        _lib.call("climb_gemm_bf16_nt", Ad, K, Wd, K, C32, N, 0, M, N, K, bd, 2, R, N, None, 0, None, 0, _st())
        assert _rel(C32[:M], ref + R.cpu().double()) < 1e-5
        Uin = _bf(torch.randn(M, N, generator=g)).to(dev)
        _lib.call("climb_gemm_bf16_nt", Ad, K, Wd, K, C16, N, 1, M, N, K, None, 3, Uin, N, None, 0, None, 0, _st())
        ur = Uin.cpu().double().requires_grad_(True)
        gelu(ur).backward(A.double() @ W.double().t())
        assert _rel(C16.float(), ur.grad) < 5e-3
        if force != 4:          # (the two-workgroup variant behind option 7 = 4 does not implement the r05 codes: they fall through to the default kernels)
            _check_saved_derivative_pair(Ad, Wd, bd, Uin, ref, A.double() @ W.double().t(), M, N, K)
    finally:
        _lib.call("climb_set_option", 7, 1)


@pytest.mark.parametrize("force", [2, 3, 4])
@pytest.mark.parametrize("N,K", [(2304, 768), (3072, 768), (768, 3072)])
def test_gemm_bf16_nt_256_race_screen_full_size(N, K, force):
    """Benchmark-size launches (M = 12288: every CU busy, several rounds) of the 256 x 256 kernel, repeated: its LDS-DMA /
    barrier placement is a hand-counted schedule, and an early fragment read would show as rare wrong tiles under load.  Both
    kernels accumulate k in the same order with the same MFMA, so the fp32 result must equal the 128 x 128 two-barrier kernel's
    BIT FOR BIT, every time."""
    from climb_amd import _lib
    dev = _dev()
    M = 12288
    g = torch.Generator(device=dev).manual_seed(N + K)
    Ad = torch.randn(M, K, device=dev, generator=g).to(_h16())
    Wd = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(_h16())
    ref = torch.empty(M, N, device=dev)
    try:
        _lib.call("climb_set_option", 7, 0)
        _lib.call("climb_set_option", 5, 0)
        _lib.call("climb_gemm_bf16_nt", Ad, K, Wd, K, ref, N, 0, M, N, K, None, 0, None, 0, None, 0, None, 0, _st())
        # spot-check the reference kernel itself on a block of rows against float64
        rows = torch.arange(0, M, 97, device=dev)
        r64 = Ad[rows].double() @ Wd.double().t()
        assert _rel(ref[rows], r64) < 1e-5
        _lib.call("climb_set_option", 7, force)
        out = torch.empty(M, N, device=dev)
        for it in range(6):
            # the tile -> (XCD, supertile) map must stay a bijection for every supertile height: automatic (one XCD's share: 6), the old 8, an odd one
            _lib.call("climb_set_option", 8, 256 * (0, 8, 5)[it % 3])
            out.fill_(float("nan"))
            _lib.call("climb_gemm_bf16_nt", Ad, K, Wd, K, out, N, 0, M, N, K, None, 0, None, 0, None, 0, None, 0, _st())
            bad = int((out != ref).sum())
            assert bad == 0, f"iteration {it}: {bad} elements differ from the two-barrier kernel"
    finally:
        _lib.call("climb_set_option", 8, 0)
        _lib.call("climb_set_option", 7, 1)
        _lib.call("climb_set_option", 5, 1)


@pytest.mark.parametrize("M,N,K", [(1536, 192, 640), (1536, 384, 704), (3072, 768, 1536), (12288, 768, 768), (12288, 2304, 768), (12288, 3072, 768), (12288, 768, 3072)])
def test_gemm_bf16_nt4_two_accumulator_sets_bit_identical(M, N, K):
    """r04: the 192 x 192 / four-wave / two-accumulator-set kernel (gemm_bf16_nt4.hip; option 17: ON by default for every epilogue but GELU, = 3 for all) drains tile i inside
    tile i + 1's k-loop: a hand-placed schedule of LDS-DMA pieces, asm operand loads, one counted wait and one barrier per k-tile.  It accumulates k in
    the order of every other NT kernel and applies the same fp32 epilogue arithmetic, so EVERY output of every epilogue must equal the 8-wave kernels'
    bit for bit -- at the minimum k-tile count (10), an odd one, one tile per workgroup (single round: everything in the exposed final drain), several
    tiles per workgroup (the overlapped drain), with and without a bias -- and stay so when repeated under load (NaN-filled outputs: a block that is
    never stored, or stored early, shows).  The 8-wave reference is itself pinned to float64 and to the two-barrier kernel by the tests above."""
    from climb_amd import _lib
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    A = torch.randn(M, K, device=dev, generator=g).to(_h16())
    W = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(_h16())
    bias = torch.randn(N, device=dev, generator=g)
    R = torch.randn(M, N, device=dev, generator=g)
    Uin = torch.randn(M, N, device=dev, generator=g).to(_h16())

    def run(cdt, epi, aux, b):
        C = torch.full((M, N), float("nan"), device=dev, dtype=_h16() if cdt else torch.float32)
        U = torch.full((M, N), float("nan"), device=dev, dtype=_h16()) if epi in (1, 8) else None
        _lib.call("climb_gemm_bf16_nt", A, K, W, K, C, N, cdt, M, N, K, b, epi, aux, N, U, N, None, 0, _st())
        return [C] + ([U] if U is not None else [])
    try:
        rows = torch.arange(0, M, 61, device=dev)
        r64 = A[rows].double() @ W.double().t() + bias.double()
        for name, cdt, epi, aux, b in [("f32", 0, 0, None, bias), ("h16", 1, 0, None, bias), ("h16 no bias", 1, 0, None, None), ("gelu", 1, 1, None, bias),
                                       ("resid", 0, 2, R, bias), ("dgelu", 1, 3, Uin, None), ("gelu + saved derivative", 1, 8, None, bias),
                                       ("x saved derivative", 1, 9, Uin, None)]:
            _lib.call("climb_set_option", 17, 0)
            ref = run(cdt, epi, aux, b)
            if name == "f32":
                assert _rel(ref[0][rows], r64) < 1e-5
            _lib.call("climb_set_option", 17, 3)          # every epilogue, GELU included (the default leaves that one to the 8-wave kernel)
            for it in range(4):
                got = run(cdt, epi, aux, b)
                for r_, g_ in zip(ref, got):
                    assert not bool(torch.isnan(g_.float()).any()), f"{name}, iteration {it}: elements never stored"
                    bad = int((g_ != r_).sum())
                    assert bad == 0, f"{name}, iteration {it}: {bad} elements differ from the 8-wave kernel"
    finally:
        _lib.call("climb_set_option", 17, 1)


def test_gemm_bf16_nt4_declines_what_it_does_not_take():
    """Shapes outside its contract (ragged tiles, fewer than 10 k-tiles, an M-tile count the 8 XCDs do not share evenly) fall through to the 8-wave kernels:
    same call, right answer."""
    from climb_amd import _lib
    dev = _dev()
    for M, N, K in [(1536, 192, 576), (1344, 192, 768), (1536, 200, 768), (200, 192, 768)]:
        g = torch.Generator().manual_seed(M + N + K)
        A, W = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(N, K, generator=g) * 0.05)
        ref = A.double() @ W.double().t()
        C = torch.empty(M, N, device=dev)
        _lib.call("climb_gemm_bf16_nt", A.to(dev), K, W.to(dev), K, C, N, 0, M, N, K, None, 0, None, 0, None, 0, None, 0, _st())
        assert _rel(C, ref) < 1e-5


@pytest.mark.parametrize("K", [1536, 2304, 3072])
@pytest.mark.parametrize("epi", ["none_16bit", "resid_fp32"])
def test_gemm_bf16_nt_split_along_k_balancing(K, epi):
    """r03 (option 14, OFF by default: measured slower): the 192-tile NT GEMMs (12288 x 768) on all 256 CUs -- four workgroups share three tiles along K and hand partial accumulator
    tiles over through the registered scratch (system-scope stores, flags).  Against the one-workgroup-per-tile kernel (itself pinned to
    float64 and to the two-barrier kernel bit for bit above): equal up to the different summation order of the split; and, repeated under
    load 25 times, BIT-IDENTICAL to its own first result every time (the hand-over is a fixed-order sum: any difference is a race)."""
    from climb_amd import _lib
    from climb_amd.engine import nt_workspace
    dev = _dev()
    M, N = 12288, 768
    g = torch.Generator(device=dev).manual_seed(K)
    Ad = torch.randn(M, K, device=dev, generator=g).to(_h16())
    Wd = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(_h16())
    bias = torch.randn(N, device=dev, generator=g)
    resid = torch.randn(M, N, device=dev, generator=g) if epi == "resid_fp32" else None
    cdt, e = (0, 2) if epi == "resid_fp32" else (1, 0)
    mk = lambda: torch.empty(M, N, device=dev, dtype=torch.float32 if cdt == 0 else _h16())

    def run(out):
        _lib.call("climb_gemm_bf16_nt", Ad, K, Wd, K, out, N, cdt, M, N, K, bias, e, resid, N, None, 0, None, 0, _st())
    nt_workspace(dev)
    try:
        _lib.call("climb_set_option", 17, 0)          # (r04: this is an experiment on the 8-wave kernel -- keep the four-wave kernel out)
        _lib.call("climb_set_option", 14, 0)
        ref = mk()
        run(ref)
        _lib.call("climb_set_option", 14, 1)
        first = mk()
        run(first)
        torch.cuda.synchronize()
        assert not torch.equal(first, ref) or K == 0          # (the split kernel really ran: a different summation order shows in the low bits)
        assert _rel(first.float(), ref.float()) < (2e-6 if cdt == 0 else 1e-2)
        rows = torch.arange(0, M, 131, device=dev)
        r64 = Ad[rows].double() @ Wd.double().t() + bias.double() + (resid[rows].double() if resid is not None else 0.0)
        assert _rel(first[rows].float(), r64) < (1e-5 if cdt == 0 else 1e-2)
        out = mk()
        for it in range(25):
            out.fill_(float("nan"))
            run(out)
            assert torch.equal(out, first), f"iteration {it}: {int((out != first).sum())} elements differ from the first run"
    finally:
        _lib.call("climb_set_option", 17, 1)
        _lib.call("climb_set_option", 14, 0)          # the library's default: measured slower than one workgroup per tile (see gemm_bf16_ntp.hip)


@pytest.mark.parametrize("value", [100, 1050, 3012])
def test_gemm_bf16_nt_dephased_start_is_the_same_arithmetic(value):
    """r03 (option 16, OFF by default: measured no gain): workgroup groups of a multi-round persistent NT launch hold back
    once before their first tile.  Only time moves: the up-projection + GELU (576 tiles on 256 workgroups, both outputs) stays bit-identical."""
    from climb_amd import _lib
    dev = _dev()
    M, N, K = 12288, 3072, 768
    g = torch.Generator(device=dev).manual_seed(value)
    Ad = torch.randn(M, K, device=dev, generator=g).to(_h16())
    Wd = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(_h16())
    bias = torch.randn(N, device=dev, generator=g)

    def run():
        out, pre = torch.empty(M, N, device=dev, dtype=_h16()), torch.empty(M, N, device=dev, dtype=_h16())
        _lib.call("climb_gemm_bf16_nt", Ad, K, Wd, K, out, N, 1, M, N, K, bias, 1, None, 0, pre, N, None, 0, _st())
        torch.cuda.synchronize()
        return out, pre
    try:
        _lib.call("climb_set_option", 17, 0)          # (r04: the knob belongs to the 8-wave kernel)
        ref = run()
        _lib.call("climb_set_option", 16, value)
        for _ in range(3):
            got = run()
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    finally:
        _lib.call("climb_set_option", 16, 0)
        _lib.call("climb_set_option", 17, 1)


@pytest.mark.parametrize("M,N,K", [(64, 128, 128), (384, 768, 768), (1000, 2304, 768), (300, 768, 3072), (130, 48, 768)])
def test_gemm_bf16_tn_weight_grad(M, N, K):
    from climb_amd import _lib
    dev = _dev()
    g = torch.Generator().manual_seed(M + 3 * N + K)
    dY, X = _bf(torch.randn(M, N, generator=g)), _bf(torch.randn(M, K, generator=g))
    C0 = torch.randn(N, K, generator=g)
    C = C0.to(dev).clone()
    db0 = torch.randn(N, generator=g)
    db = db0.to(dev).clone()
    _lib.call("climb_gemm_bf16_tn", dY.to(dev), N, X.to(dev), K, C, K, M, N, K, db, _st())
    ref = C0.double() + dY.double().t() @ X.double()
    assert _rel(C, ref) < 1e-5
    assert _rel(db, db0.double() + dY.double().sum(0)) < 1e-5          # fused bias gradient (dY^T . 1)


@pytest.mark.parametrize("S_pad,valid", [(32, 20), (64, 50), (96, 96), (160, 150), (192, 185), (224, 200), (288, 281), (352, 331), (416, 400), (512, 512)])
def test_attention_bf16_fwd_bwd(S_pad, valid):
    from climb_amd import _lib
    dev = _dev()
    B, heads, d = 2, 3, 64
    H = heads * d
    g = torch.Generator().manual_seed(S_pad + 1)
    qkv = _bf(torch.randn(B, S_pad, 3 * H, generator=g))
    bias = torch.zeros(B, S_pad)
    bias[:, valid:] = -3.0e38
    bias[1, 3:7] = -3.0e38
    dctx = _bf(torch.randn(B, S_pad, H, generator=g))
    dctx[:, valid:] = 0
    qr = qkv.double().requires_grad_(True)
    ref = _attn_ref(qr, bias.double().clamp(min=-1e300), heads)
    ref.backward(dctx.double())
    qd, bd, dd = qkv.to(dev).view(B * S_pad, 3 * H), bias.to(dev), dctx.to(dev).view(B * S_pad, H)
    ctx = torch.empty(B * S_pad, H, device=dev, dtype=_h16())
    lse = torch.empty(B, heads, S_pad, device=dev)
    _lib.call("climb_attn_fwd_bf16", qd, bd, ctx, lse, B, S_pad, heads, d, _st())
    e_fwd = _rel(ctx.float().view(B, S_pad, H)[:, :valid], ref.detach()[:, :valid])
    delta = torch.empty(B, heads, S_pad, device=dev)
    dqkv = torch.full((B * S_pad, 3 * H), float("nan"), device=dev, dtype=_h16())
    delta.fill_(float("nan"))           # scratch: written by the kernel's first phase
    _lib.call("climb_attn_bwd_bf16", qd, bd, dd, ctx, lse, delta, dqkv, B, S_pad, heads, d, _st())
    assert not torch.isnan(dqkv.float()).any()
    e_bwd = _rel(dqkv.float().view(B, S_pad, 3 * H), qr.grad)
    print(f"attention bf16 S_pad={S_pad}: fwd {e_fwd:.2e} bwd {e_bwd:.2e}")
    # P and dS are rounded to bf16 before the second MFMA of each product: 2^-8 relative per element, averaged by the sums
    assert e_fwd < 1e-2 and e_bwd < 2e-2


@pytest.mark.parametrize("S_pad", [96, 192, 288])
def test_attention_bf16_forward_variants_agree(S_pad):
    """one / two query blocks per wave (climb_set_option 12) are the same arithmetic in a different wave shape: identical results"""
    from climb_amd import _lib
    dev = _dev()
    B, heads, d = 3, 4, 64
    H = heads * d
    g = torch.Generator().manual_seed(S_pad)
    qkv = _bf(torch.randn(B * S_pad, 3 * H, generator=g)).to(dev)
    bias = torch.zeros(B, S_pad)
    bias[:, S_pad - 7:] = -3.0e38
    bias = bias.to(dev)
    outs = []
    try:
        for qb in (1, 2, 0):
            _lib.call("climb_set_option", 12, qb)
            ctx = torch.empty(B * S_pad, H, device=dev, dtype=_h16())
            lse = torch.empty(B, heads, S_pad, device=dev)
            _lib.call("climb_attn_fwd_bf16", qkv, bias, ctx, lse, B, S_pad, heads, d, _st())
            outs.append((ctx.float().cpu(), lse.cpu()))
    finally:
        _lib.call("climb_set_option", 12, 0)
    for c, l in outs[1:]:
        assert torch.equal(c, outs[0][0]) and torch.equal(l, outs[0][1])


@pytest.mark.parametrize("S_pad,valid", [(32, 32), (64, 41), (128, 128), (160, 131), (192, 185)])
def test_attention_bf16_backward_variants_agree(S_pad, valid):
    """The backward kernels (climb_set_option 13: 0 = one launch per phase, 1 = both phases in one launch, 3 = the single pass that computes
    S / dP / P / dS once per block pair and keeps dQ in LDS, 4 = its persistent form that walks (batch, head) items with the next item's Q / dO
    images prefetched -- the default, 2, takes 3 where S_pad <= 128 and 1 above) against the fp64 softmax backward.  0 and 1 are the same
    arithmetic (identical), 3 and 4 too; the single pass sums dQ over key blocks in another order than the phases (close, not identical).
    The persistent kernel runs on a grid of 4 workgroups here (option 20): every workgroup walks 3 - 4 items through both image buffers."""
    from climb_amd import _lib
    dev = _dev()
    B, heads, d = 3, 5, 64
    H = heads * d
    g = torch.Generator().manual_seed(7 * S_pad + valid)
    qkv = _bf(torch.randn(B, S_pad, 3 * H, generator=g))
    bias = torch.zeros(B, S_pad)
    bias[:, valid:] = -3.0e38
    bias[2, 1:4] = -3.0e38
    dctx = _bf(torch.randn(B, S_pad, H, generator=g))
    dctx[:, valid:] = 0
    qr = qkv.double().requires_grad_(True)
    _attn_ref(qr, bias.double().clamp(min=-1e300), heads).backward(dctx.double())
    qd, bd, dd = qkv.to(dev).view(B * S_pad, 3 * H), bias.to(dev), dctx.to(dev).view(B * S_pad, H)
    ctx = torch.empty(B * S_pad, H, device=dev, dtype=_h16())
    lse = torch.empty(B, heads, S_pad, device=dev)
    _lib.call("climb_attn_fwd_bf16", qd, bd, ctx, lse, B, S_pad, heads, d, _st())
    outs = {}
    try:
        _lib.call("climb_set_option", 20, 4)
        for mode in (0, 1, 3, 4):
            _lib.call("climb_set_option", 13, mode)
            delta = torch.full((B, heads, S_pad), float("nan"), device=dev)
            dqkv = torch.full((B * S_pad, 3 * H), float("nan"), device=dev, dtype=_h16())
            for _ in range(3):          # repeated launches into the same output: the LDS accumulators must not carry anything over
                _lib.call("climb_attn_bwd_bf16", qd, bd, dd, ctx, lse, delta, dqkv, B, S_pad, heads, d, _st())
            outs[mode] = dqkv.float().view(B, S_pad, 3 * H).cpu()
    finally:
        _lib.call("climb_set_option", 13, 2)
        _lib.call("climb_set_option", 20, 0)
    assert torch.equal(outs[0], outs[1])
    assert torch.equal(outs[3], outs[4])
    for mode in (1, 3, 4):
        assert not torch.isnan(outs[mode]).any()
        e = [_rel(outs[mode][..., k * H:(k + 1) * H], qr.grad[..., k * H:(k + 1) * H]) for k in range(3)]
        print(f"attention bwd mode {mode} S_pad={S_pad}: dq {e[0]:.2e} dk {e[1]:.2e} dv {e[2]:.2e}")
        assert max(e) < 2e-2
    assert _rel(outs[3], outs[1]) < 1e-2


@pytest.mark.parametrize("case", ["soft_bias", "peaked", "masked_row"])
def test_attention_bf16_edge_inputs(case):
    """The bf16 attention's lazy running maximum and folded bias under inputs the step never produces: a general additive bias (not only
    0 / -3e38), scores large enough that every block moves the maximum, and a row whose keys are all masked (defined as 0, not NaN)."""
    from climb_amd import _lib
    dev = _dev()
    B, heads, d, S_pad = 2, 2, 64, 192
    H = heads * d
    g = torch.Generator().manual_seed({"soft_bias": 1, "peaked": 2, "masked_row": 3}[case])
    qkv = torch.randn(B, S_pad, 3 * H, generator=g)
    bias = torch.zeros(B, S_pad)
    if case == "soft_bias":
        bias = torch.randn(B, S_pad, generator=g) * 3.0
    elif case == "peaked":
        qkv[..., :2 * H] *= 6.0                    # scores ~ N(0, 36^2): the maximum grows by far more than 2^8 between blocks
    else:
        bias[1, :] = -3.0e38                       # batch 1: every key masked
    qkv = _bf(qkv)
    dctx = _bf(torch.randn(B, S_pad, H, generator=g))
    qd, bd, dd = qkv.to(dev).view(B * S_pad, 3 * H), bias.to(dev), dctx.to(dev).view(B * S_pad, H)
    ctx = torch.empty(B * S_pad, H, device=dev, dtype=_h16())
    lse = torch.empty(B, heads, S_pad, device=dev)
    delta = torch.empty(B, heads, S_pad, device=dev)
    dqkv = torch.empty(B * S_pad, 3 * H, device=dev, dtype=_h16())
    _lib.call("climb_attn_fwd_bf16", qd, bd, ctx, lse, B, S_pad, heads, d, _st())
    _lib.call("climb_attn_bwd_bf16", qd, bd, dd, ctx, lse, delta, dqkv, B, S_pad, heads, d, _st())
    c, dq = ctx.float().view(B, S_pad, H).cpu(), dqkv.float().view(B, S_pad, 3 * H).cpu()
    assert torch.isfinite(c).all() and torch.isfinite(dq).all() and torch.isfinite(lse).all()
    nb = 1 if case == "masked_row" else B          # rows to compare against the fp64 softmax
    qr = qkv[:nb].double().requires_grad_(True)
    ref = _attn_ref(qr, bias[:nb].double(), heads)
    ref.backward(dctx[:nb].double())
    e_fwd, e_bwd = _rel(c[:nb], ref.detach()), _rel(dq[:nb], qr.grad)
    print(f"attention bf16 {case}: fwd {e_fwd:.2e} bwd {e_bwd:.2e}")
    assert e_fwd < 1e-2 and e_bwd < (4e-2 if case == "peaked" else 2e-2)
    if case == "masked_row":
        assert (c[1] == 0).all() and (dq[1] == 0).all()


@pytest.mark.parametrize("M,r", [(12288, 48), (77, 48), (1000, 64), (96, 16)])
def test_adapter_forward_fused(M, r):
    """The one-launch Houlsby adapter forward against float64 of the same 16-bit-rounded operands: z, s = silu(z), out = resid + y + s Wu^T + bu."""
    from climb_amd import _lib
    dev = _dev()
    H = 768
    g = torch.Generator().manual_seed(M + r)
    y = _bf(torch.randn(M, H, generator=g))
    resid = torch.randn(M, H, generator=g)
    wd, wu = _bf(torch.randn(r, H, generator=g) * 0.05), _bf(torch.randn(H, r, generator=g) * 0.05)
    bd, bu = torch.randn(r, generator=g) * 0.1, torch.randn(H, generator=g) * 0.1
    z = torch.full((M, r), float("nan"), device=dev, dtype=_h16())
    s = torch.full((M, r), float("nan"), device=dev, dtype=_h16())
    out = torch.full((M, H), float("nan"), device=dev)
    _lib.call("climb_adapter_fwd_bf16", y.to(dev), H, resid.to(dev), H, wd.to(dev), bd.to(dev), wu.to(dev), bu.to(dev), z, s, r, out, H, M, H, r, _st())
    zr = y.double() @ wd.double().t() + bd.double()
    sr = zr * torch.sigmoid(zr)
    s16 = sr.to(_h16()).double()                      # the up-projection consumes the 16-bit s
    outr = resid.double() + y.double() + s16 @ wu.double().t() + bu.double()
    assert _rel(z.float(), zr) < 6e-3 and _rel(s.float(), sr) < 6e-3
    assert _rel(out, outr) < 1e-3        # s is rounded to 16 bits from the kernel's own fp32 z: one-ulp differences against the float64 s reach out at ~3e-4
    if r == 48 and M == 12288:      # the two-GEMM path it replaces computes the same thing
        z2 = torch.empty_like(z); s2 = torch.empty_like(s); out2 = torch.empty_like(out)
        _lib.call("climb_gemm_bf16_nt", y.to(dev), H, wd.to(dev), H, s2, r, 1, M, r, H, bd.to(dev), 5, None, 0, z2, r, None, 0, _st())
        _lib.call("climb_gemm_bf16_nt", s2, r, wu.to(dev), r, out2, H, 0, M, H, r, bu.to(dev), 7, resid.to(dev), H, None, 0, y.to(dev), H, _st())
        assert _rel(out, out2.double().cpu()) < 1e-3 and _rel(s.float(), s2.float().double().cpu()) < 6e-3


@pytest.mark.parametrize("M,r", [(12288, 48), (100, 48), (64, 16), (517, 64)])
def test_adapter_backward_fused(M, r):
    """r03: the input-gradient half of the adapter backward in one launch, dz = (dout Wu) * silu'(z), dy = dres + dz Wd, against float64 of the
    same 16-bit-rounded operands (the kernel reads the TRANSPOSED weight shadows)."""
    from climb_amd import _lib
    dev = _dev()
    H = 768
    g = torch.Generator().manual_seed(3 * M + r)
    dout = _bf(torch.randn(M, H, generator=g))
    dres = torch.randn(M, H, generator=g)
    wd, wu = _bf(torch.randn(r, H, generator=g) * 0.05), _bf(torch.randn(H, r, generator=g) * 0.05)
    zin = _bf(torch.randn(M, r, generator=g))
    dz = torch.full((M, r), float("nan"), device=dev, dtype=_h16())
    dy = torch.full((M, H), float("nan"), device=dev, dtype=_h16())
    _lib.call("climb_adapter_bwd_bf16", dout.to(dev), H, dres.to(dev), H, wu.t().contiguous().to(dev), wd.t().contiguous().to(dev), zin.to(dev), dz, r,
              dy, H, M, H, r, _st())
    t = dout.double() @ wu.double()
    zd = zin.double()
    sg = torch.sigmoid(zd)
    dzr = t * (sg * (1.0 + zd * (1.0 - sg)))
    dz16 = dzr.to(_h16()).double()
    dyr = dres.double() + dz16 @ wd.double()
    assert _rel(dz.float(), dzr) < 6e-3
    assert _rel(dy.float(), dyr) < 6e-3          # 16-bit output


def test_weight_shadow_cast_and_batched_transpose():
    from climb_amd import _lib
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2 * 96 * 160, generator=g)
    xd = x.to(dev)
    sh = torch.empty_like(xd, dtype=_h16())
    _lib.call("climb_cast_bf16", xd, sh, x.numel(), _st())
    assert torch.equal(sh.cpu(), x.to(_h16()))
    out = torch.empty_like(sh)
    table = torch.tensor([[0, 0, 96, 160], [96 * 160, 96 * 160, 160, 96]], dtype=torch.int64, device=dev)
    _lib.call("climb_transpose_bf16_batched", sh, out, table, 2, 4, _st())
    a = sh[:96 * 160].view(96, 160).t().contiguous().view(-1)
    b = sh[96 * 160:].view(160, 96).t().contiguous().view(-1)
    assert torch.equal(out.cpu(), torch.cat([a, b]).cpu())


# ------------------------------------------------------------------------------------------------ error behaviour of the C ABI
def test_abi_rejects_bad_arguments_with_codes_not_crashes():
    """SURVEY.md §8(b) "Errors": every entry returns an int; invalid shapes / missing operands come back as a library code (turned into
    RuntimeError by the binding) instead of launching, throwing across the ABI or exiting."""
    from climb_amd import _lib
    dev = _dev()
    a = torch.zeros(64, 64, device=dev, dtype=_h16())
    c = torch.zeros(64, 64, device=dev)
    bad = [
        ("climb_gemm_bf16_nt", (a, 64, a, 64, c, 64, 0, 0, 64, 64, None, 0, None, 0, None, 0, None, 0, _st())),       # M = 0
        ("climb_gemm_bf16_nt", (a, 64, a, 64, c, 64, 0, 64, 64, 60, None, 0, None, 0, None, 0, None, 0, _st())),      # K % 8 != 0
        ("climb_gemm_bf16_nt", (a, 64, a, 64, c, 64, 0, 64, 64, 64, None, 2, None, 0, None, 0, None, 0, _st())),      # residual epilogue, no residual
        ("climb_gemm_bf16_nt", (a, 64, a, 64, c, 64, 5, 64, 64, 64, None, 0, None, 0, None, 0, None, 0, _st())),      # unknown output dtype
        ("climb_gemm_bf16_tn", (a, 64, a, 64, c, 64, 64, 60, 64, None, _st())),                                         # N % 8 != 0
        ("climb_gemm_f32", (c, 64, 1, c, 64, 1, c, 64, 64, 64, 64, None, 1, None, 0, None, 0, 0.0, None, 0, 0, _st())),  # GELU without aux_out
        ("climb_attn_fwd_bf16", (a, c, a, c, 1, 48, 1, 64, _st())),                                                      # S_pad % 32 != 0
        ("climb_attn_fwd_bf16", (a, c, a, c, 1, 64, 1, 32, _st())),                                                      # head_dim != 64
        ("climb_layernorm_fwd", (c, 64, c, c, 1e-5, c, 64, 0, c, c, 64, 66, _st())),                                     # C % 4 != 0
        ("climb_image_resample", (a, a, a, c, c, 0, 1, _st())),                                                          # no images
        ("climb_cast_bf16", (c, a, 6, _st())),                                                                           # n % 4 != 0
    ]
    for name, args in bad:
        with pytest.raises(RuntimeError, match=name):
            _lib.call(name, *args)
    torch.cuda.synchronize()          # nothing was launched, the context is healthy
    assert _lib.load().climb_error_string(-1).decode().startswith("climb:")
    assert _lib.load().climb_error_string(0).decode() == "ok"
    assert _lib.load().climb_arch().decode() == "gfx950" and _lib.load().climb_version() >= 100
