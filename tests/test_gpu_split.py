"""Split-operand arithmetic (r06, csrc/split.hip): the passes that produce (hi, lo) 16-bit plane pairs and the three-phase GEMMs over them, each
against float64 on the CPU.  Two bars per GEMM: against float64 of the SAME planes (what the kernel was given: only the fp32 accumulation differs,
<= 1e-5), and against float64 of the ORIGINAL fp32 operands (what the mode promises: the lo.lo term and the planes' own rounding, <= 4e-5 of the
output's scale -- 25x inside north_star's 1e-3)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from climb_amd import _lib
    if _lib.h16() != "bf16":
        pytest.skip("split operands are built on the bf16 library")
    return torch.device("cuda:0")


def _st():
    return torch.cuda.current_stream().cuda_stream


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _planes(x):
    """the oracle's split: hi = rn_bf16(x), lo = rn_bf16(x - hi)"""
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return hi, lo


def _split_dev(x):
    """[2][M][C] planes of an fp32 device matrix, through the library's own pass"""
    from climb_amd import _lib
    M, C = x.shape
    out = torch.empty((2, M, C), dtype=torch.bfloat16, device=x.device)
    _lib.call("climb_split_f32", x, C, out, C, M * C, M, C, 0, None, 0, _st())
    return out


def gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def dgelu(x):
    return 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2.0 * math.pi)


@pytest.mark.parametrize("M,C", [(1, 4096), (37, 768), (200, 3072)])
def test_split_pass_planes_and_activations(M, C):
    from climb_amd import _lib
    dev = _dev()
    g = torch.Generator().manual_seed(M + C)
    x = torch.randn(M, C, generator=g) * torch.logspace(-6, 3, C)[None, :]          # nine decades: the lo plane keeps bf16's range
    v = torch.randn(M, C, generator=g)
    u = torch.randn(M, C, generator=g) * 2.0
    xd, vd, ud = x.to(dev), v.to(dev), u.to(dev)
    out = _split_dev(xd)
    hi, lo = _planes(x)
    assert torch.equal(out[0].cpu(), hi) and torch.equal(out[1].cpu(), lo)          # bit-exact: two hardware round-to-nearest-even converts
    rel = ((out[0].double() + out[1].double()).cpu() - x.double()).abs() / x.double().abs().clamp_min(1e-30)
    assert float(rel.max()) < 2.0 ** -16                                            # element-wise: 16 significant bits
    o1 = torch.empty_like(out)
    _lib.call("climb_split_f32", ud, C, o1, C, M * C, M, C, 1, None, 0, _st())
    assert _rel(o1[0].double() + o1[1].double(), gelu(u.double())) < 1e-5
    o2 = torch.empty_like(out)
    _lib.call("climb_split_f32", vd, C, o2, C, M * C, M, C, 2, ud, C, _st())
    assert _rel(o2[0].double() + o2[1].double(), v.double() * dgelu(u.double())) < 1e-5


@pytest.mark.parametrize("M,N,K,epi", [(1536, 192, 256, 0), (1536, 768, 768, 2), (3072, 2304, 768, 0), (12288, 768, 3072, 2), (12288, 3072, 768, 0),
                                       (384, 768, 768, 2), (200, 2304, 768, 0), (77, 48, 128, 0), (384, 768, 3072, 2)])
def test_gemm_split_nt(M, N, K, epi):
    """forward / input-gradient GEMM on split operands: the four-wave persistent kernel where the shape tiles into 192 x 192 (the first five), the
    128 x 128 kernel otherwise (ragged M, small N)"""
    from climb_amd import _lib
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(M + 3 * N + K)
    A = torch.randn(M, K, device=dev, generator=g)
    W = torch.randn(N, K, device=dev, generator=g) * 0.05
    bias = torch.randn(N, device=dev, generator=g)
    res = torch.randn(M, N, device=dev, generator=g)
    As, Ws = _split_dev(A), _split_dev(W)
    C = torch.full((M, N), float("nan"), device=dev)
    _lib.call("climb_gemm_split_nt", As, K, M * K, Ws, K, N * K, C, N, M, N, K, bias, epi, res if epi == 2 else None, N, _st())
    torch.cuda.synchronize()
    extra = bias.double().cpu() + (res.double().cpu() if epi == 2 else 0.0)
    Ah, Al, Wh, Wl = (t.double().cpu() for t in (As[0], As[1], Ws[0], Ws[1]))
    same_planes = Ah @ Wh.t() + Ah @ Wl.t() + Al @ Wh.t() + extra
    assert _rel(C, same_planes) < 1e-5
    exact = A.double().cpu() @ W.double().cpu().t() + extra
    assert _rel(C, exact) < 4e-5
    # the lo planes at an arbitrary distance (the weight shadow's planes are a whole parameter buffer apart)
    far = torch.zeros(2 * N * K + 4096, dtype=torch.bfloat16, device=dev)
    far[:N * K].copy_(Ws[0].reshape(-1))
    far[N * K + 4096:].copy_(Ws[1].reshape(-1))
    C2 = torch.empty_like(C)
    _lib.call("climb_gemm_split_nt", As, K, M * K, far, K, N * K + 4096, C2, N, M, N, K, bias, epi, res if epi == 2 else None, N, _st())
    assert torch.equal(C, C2)


@pytest.mark.parametrize("M,N,K", [(384, 768, 768), (1000, 2304, 768), (320, 768, 3072), (2048, 256, 512)])
def test_gemm_split_tn_three_launches(M, N, K):
    from climb_amd import _lib
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(M + N)
    dY = torch.randn(M, N, device=dev, generator=g)
    X = torch.randn(M, K, device=dev, generator=g)
    C0 = torch.randn(N, K, device=dev, generator=g)
    b0 = torch.randn(N, device=dev, generator=g)
    dYs, Xs = _split_dev(dY), _split_dev(X)
    C, db = C0.clone(), b0.clone()
    _lib.call("climb_gemm_split_tn", dYs, N, M * N, Xs, K, M * K, C, K, M, N, K, db, _st())
    torch.cuda.synchronize()
    exact = C0.double().cpu() + dY.double().cpu().t() @ X.double().cpu()
    assert _rel(C, exact) < 4e-5
    assert _rel(db, b0.double().cpu() + dY.double().cpu().sum(0)) < 2e-5


@pytest.mark.parametrize("nwg", [256, 8, 24])
def test_gemm_split_tn_grouped_launch(nwg):
    """the grouped weight-gradient launch over split operands: three phases over the stacked planes, whole tiles and stream-K shares that cut
    through phase boundaries, bias gradients = column sums of hi + lo (the repeated hi phase left out)"""
    from climb_amd import _lib
    dev = _dev()
    shapes = [(2048, 768, 768, True), (2048, 256, 512, False), (1024, 512, 256, True), (3072, 256, 256, False), (1152, 768, 256, True)]
    g = torch.Generator(device=dev).manual_seed(11 + nwg)
    ops = []
    for M, N, K, bias in shapes:
        dY, X = torch.randn(M, N, device=dev, generator=g), torch.randn(M, K, device=dev, generator=g)
        ops.append((dY, X, _split_dev(dY), _split_dev(X), torch.randn(N, K, device=dev, generator=g), torch.randn(N, device=dev, generator=g) if bias else None))
    rec = np.zeros(len(shapes), dtype=[("A", "<u8"), ("B", "<u8"), ("C", "<u8"), ("dbias", "<u8"), ("lda", "<i8"), ("ldb", "<i8"), ("ldc", "<i8"),
                                       ("M", "<i4"), ("N", "<i4"), ("K", "<i4"), ("reserved", "<i4")])
    Cs = [o[4].clone() for o in ops]
    dbs = [o[5].clone() if o[5] is not None else None for o in ops]
    for r, (M, N, K, _), o, C, db in zip(rec, shapes, ops, Cs, dbs):
        r["A"], r["B"], r["C"], r["dbias"] = o[2].data_ptr(), o[3].data_ptr(), C.data_ptr(), (db.data_ptr() if db is not None else 0)
        r["lda"], r["ldb"], r["ldc"], r["M"], r["N"], r["K"], r["reserved"] = N, K, K, 3 * M, N, K, M // 64
    Ms, Ns, Ks = (np.ascontiguousarray(rec[f], dtype=np.int32) for f in ("M", "N", "K"))
    cap = int(sum(((n + 255) // 256) * ((k + 255) // 256) for n, k in zip(Ns, Ks))) + 2 * nwg + 1
    items, first = np.zeros((cap, 8), dtype=np.int32), np.zeros(nwg + 1, dtype=np.int32)
    n = _lib.load().climb_tn_grouped_plan(len(shapes), Ms.ctypes.data, Ns.ctypes.data, Ks.ctypes.data, nwg, items.ctypes.data, cap, first.ctypes.data)
    assert n > 0
    d_rec = torch.from_numpy(rec.view(np.uint8).copy()).to(dev)
    d_items, d_first = torch.from_numpy(items[:n].copy()).to(dev), torch.from_numpy(first).to(dev)
    _lib.call("climb_gemm_split_tn_grouped", d_rec, d_items, d_first, nwg, _st())
    torch.cuda.synchronize()
    for (dY, X, _, _, C0, b0), C, db in zip(ops, Cs, dbs):
        exact = C0.double().cpu() + dY.double().cpu().t() @ X.double().cpu()
        assert _rel(C, exact) < 4e-5
        if b0 is not None:
            assert _rel(db, b0.double().cpu() + dY.double().cpu().sum(0)) < 2e-5


@pytest.mark.parametrize("C", [768, 1536])
def test_layernorm_split_outputs(C):
    """LayerNorm forward writing its output, and the backward writing the cast of its result, as split planes: the planes of the fp32 kernels' results"""
    from climb_amd import _lib
    dev = _dev()
    M = 100
    g = torch.Generator(device=dev).manual_seed(C)
    x = torch.randn(M, C, device=dev, generator=g) * 3.0 + 0.5
    gamma, beta = torch.randn(C, device=dev, generator=g), torch.randn(C, device=dev, generator=g)
    y32 = torch.empty(M, C, device=dev)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    _lib.call("climb_layernorm_fwd", x, C, gamma, beta, 1e-12, y32, C, 0, mean, rstd, M, C, _st())
    ys = torch.empty((2, M, C), dtype=torch.bfloat16, device=dev)
    _lib.call("climb_layernorm_fwd", x, C, gamma, beta, 1e-12, ys, C, 2, mean, rstd, M, C, _st())
    hi, lo = _planes(y32.cpu())
    assert torch.equal(ys[0].cpu(), hi) and torch.equal(ys[1].cpu(), lo)
    dy = torch.randn(M, C, device=dev, generator=g)
    dres = torch.randn(M, C, device=dev, generator=g)
    lnb = _lib.query("climb_layernorm_bwd_rows_per_block")
    part = torch.empty(((M + lnb - 1) // lnb) * 3 * C, device=dev)
    o32 = torch.empty(M, C, device=dev)
    _lib.call("climb_layernorm_bwd", dy, C, 0, x, C, mean, rstd, gamma, dres, C, o32, C, None, 0, part, M, C, _st())
    o2, cs = torch.empty(M, C, device=dev), torch.empty((2, M, C), dtype=torch.bfloat16, device=dev)
    _lib.call("climb_layernorm_bwd", dy, C, 2, x, C, mean, rstd, gamma, dres, C, o2, C, cs, C, part, M, C, _st())
    assert torch.equal(o2, o32)
    hi, lo = _planes(o32.cpu())
    assert torch.equal(cs[0].cpu(), hi) and torch.equal(cs[1].cpu(), lo)


def test_split_abi_rejects_bad_arguments():
    from climb_amd import _lib
    dev = _dev()
    lib = _lib.load()
    x = torch.zeros(8, 64, device=dev)
    y = torch.zeros((2, 8, 64), dtype=torch.bfloat16, device=dev)
    c = torch.zeros(8, 8, device=dev)
    assert lib.climb_split_f32(x.data_ptr(), 64, y.data_ptr(), 64, 512, 8, 63, 0, None, 0, _st()) == -1          # C % 4
    assert lib.climb_split_f32(x.data_ptr(), 64, y.data_ptr(), 64, 512, 8, 64, 2, None, 0, _st()) == -1          # mode 2 without aux
    assert lib.climb_gemm_split_nt(y.data_ptr(), 64, 512, y.data_ptr(), 64, 512, c.data_ptr(), 8, 8, 8, 48, None, 0, None, 0, _st()) == -1      # K % 64
    assert lib.climb_gemm_split_nt(y.data_ptr(), 64, 512, y.data_ptr(), 64, 512, c.data_ptr(), 8, 8, 8, 64, None, 1, None, 0, _st()) == -1      # epilogue
    assert lib.climb_gemm_split_nt(y.data_ptr(), 64, 512, y.data_ptr(), 64, 512, c.data_ptr(), 8, 8, 8, 64, None, 2, None, 0, _st()) == -1      # residual missing


def _attn_ref(qkv, bias, heads):
    B, S, H3 = qkv.shape
    H = H3 // 3
    d = H // heads
    q, k, v = (qkv[..., i * H:(i + 1) * H].reshape(B, S, heads, d).transpose(1, 2) for i in range(3))
    s = q @ k.transpose(-1, -2) / math.sqrt(d) + bias[:, None, None, :]
    return (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, S, H)


@pytest.mark.parametrize("S_pad,valid", [(32, 20), (64, 50), (160, 150), (192, 185), (224, 200), (288, 281)])
def test_attention_split_fwd_bwd(S_pad, valid):
    """the split-operand attention against float64 autograd of the same fp32 inputs: forward ctx (fp32 and planes) and lse, backward d(qkv) (fp32 and
    planes); 4 / 6 / 8 waves per workgroup and the two-chunk path (S_pad > 192) are all among the shapes"""
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
    M = B * S_pad
    qd, bd, dd = qkv.to(dev).view(M, 3 * H), bias.to(dev), dctx.to(dev).view(M, H)
    ctx = torch.full((M, H), float("nan"), device=dev)
    ctx_s = torch.empty((2, M, H), dtype=torch.bfloat16, device=dev)
    lse = torch.empty(B, heads, S_pad, device=dev)
    _lib.call("climb_attn_fwd_split", qd, bd, ctx, ctx_s, M * H, lse, B, S_pad, heads, d, _st())
    assert _rel(ctx.view(B, S_pad, H)[:, :valid], ref.detach()[:, :valid]) < 3e-5
    hi, lo = _planes(ctx.cpu())
    assert torch.equal(ctx_s[0].cpu(), hi) and torch.equal(ctx_s[1].cpu(), lo)
    # against the exact-fp32 kernels' statistics
    ctx32, lse32 = torch.empty(M, H, device=dev), torch.empty(B, heads, S_pad, device=dev)
    _lib.call("climb_attn_fwd_f32", qd, bd, ctx32, lse32, B, S_pad, heads, d, _st())
    assert float((lse[..., :valid] - lse32[..., :valid]).abs().max()) < 2e-4
    delta = torch.empty(B, heads, S_pad, device=dev)
    dqkv = torch.full((M, 3 * H), float("nan"), device=dev)
    dq_s = torch.empty((2, M, 3 * H), dtype=torch.bfloat16, device=dev)
    _lib.call("climb_attn_delta", dd, ctx, 0, delta, B, S_pad, heads, _st())
    _lib.call("climb_attn_bwd_split", qd, bd, dd, lse, delta, dqkv, dq_s, M * 3 * H, B, S_pad, heads, d, _st())
    assert not torch.isnan(dqkv).any()
    assert _rel(dqkv.view(B, S_pad, 3 * H), qr.grad) < 4e-5
    hi, lo = _planes(dqkv.cpu())
    assert torch.equal(dq_s[0].cpu(), hi) and torch.equal(dq_s[1].cpu(), lo)
    # planes only (what the engine asks for)
    dq2 = torch.empty_like(dq_s)
    _lib.call("climb_attn_bwd_split", qd, bd, dd, lse, delta, None, dq2, M * 3 * H, B, S_pad, heads, d, _st())
    assert torch.equal(dq2, dq_s)


# ------------------------------------------------------------------------------------------------ the bf16x3 ENGINE on every step path, against the fp32 mode
def _both_modes(tasks=("vqa", "nlvr2")):
    from tests.test_gpu_parity import make_model
    a, _ = make_model(list(tasks), 42, precision="fp32")
    b, _ = make_model(list(tasks), 42, precision="bf16x3")
    return a, b


def _grad_err(ga, gb):
    worst = (0.0, None)
    for n in ga:
        if n.endswith("attention.key.bias"):          # identically zero in exact arithmetic: rounding noise on both sides
            continue
        assert n in gb, n
        e = float((gb[n].double() - ga[n].double()).abs().max() / (ga[n].double().abs().max() + 1e-30))
        worst = max(worst, (e, n))
    assert set(ga) == set(gb)
    return worst


@pytest.mark.parametrize("path", ["autograd", "ewc", "frozen9", "accumulate", "frozen_encoder", "hipgraph", "optimizer_steps", "ewc_named_optimizer"])
def test_bf16x3_engine_on_every_step_path_against_the_fp32_mode(path):
    """The split-operand mode shares the fp32 mode's host code; this runs it down the paths the fixture tests do not take -- the reference-style
    autograd path (model(...) -> torch loss -> loss.backward()), the EWC penalty, a frozen prefix (the backward stops at layer 9), gradient
    accumulation without zero_grad (the Fisher pass's pattern), a frozen encoder, the captured hipGraph step and three AdamW steps (plane refresh
    after every update) -- and compares every gradient (or parameter) with the exact-fp32 mode's: <= 3e-4 of each tensor's scale."""
    import types
    from oracle import vilt_oracle as vo
    from tests.test_gpu_parity import _ewc_state, enc_to_inputs, grads_of
    _dev()
    ma, mb = _both_modes()
    enc = vo.synthetic_encodings(3, seed=5, ragged_text=True)
    images, texts = enc_to_inputs(enc)
    target = vo.synthetic_vqa_targets(3, seed=5)
    out = []
    for model in (ma, mb):
        model.train()
        if path == "autograd":
            model.zero_grad()
            o = model(task_key="vqa", images=images, texts=texts)
            loss = torch.nn.BCEWithLogitsLoss(reduction="mean")(o[1], target.to(o[1].device)) * target.shape[1]
            loss.backward()
            out.append((float(loss), o[1].detach().float().cpu(), grads_of(model)))
        elif path == "ewc":
            from climb_amd.cl_algorithms import EWC
            ewc = EWC(types.SimpleNamespace(ewc_fisher_sample_percentage=0.01, ewc_loss_weight=100.0))
            P = {n: p.detach().cpu() for n, p in model.named_parameters()}
            fisher, star = _ewc_state(P, 5)
            ewc.set_task_state("nlvr2", model, fisher, star)
            loss, (_, logits), task, eloss = model.fused_forward_backward("vqa", images, texts, target, ewc)
            out.append((float(loss) + float(eloss), logits.detach().float().cpu(), grads_of(model)))
        elif path == "frozen9":
            model.get_encoder().freeze_bottom_k_layers(9)
            loss, (_, logits), _, _ = model.fused_forward_backward("vqa", images, texts, target)
            G = grads_of(model)
            assert all(".layer.8." not in n and "embeddings" not in n for n in G)
            out.append((float(loss), logits.detach().float().cpu(), G))
        elif path == "frozen_encoder":
            model.get_encoder().freeze_all_weights()
            loss, (_, logits), _, _ = model.fused_forward_backward("vqa", images, texts, target)
            G = grads_of(model)
            assert all(n.startswith("task_layer.") for n in G) and G
            out.append((float(loss), logits.detach().float().cpu(), G))
        elif path == "accumulate":
            model.fused_forward_backward("vqa", images, texts, target)
            loss, (_, logits), _, _ = model.fused_forward_backward("vqa", images, texts, target)          # no zero_grad: sums, like .grad
            out.append((float(loss), logits.detach().float().cpu(), grads_of(model)))
        elif path == "hipgraph":
            opt = model.create_optimizer({"lr": 1e-3, "weight_decay": 1e-2, "adam_epsilon": 1e-8})
            for _ in range(2):          # (the first call captures, the second replays)
                opt.zero_grad()
                loss, (_, logits), _, _ = model.graphed_forward_backward("vqa", images, texts, target)
            torch.cuda.synchronize()
            loss, logits = loss.clone(), logits.clone()          # (the graph's static outputs)
            out.append((float(loss), logits.detach().float().cpu(), grads_of(model)))
        elif path == "ewc_named_optimizer":
            # the trainers' call: the optimizer is named, the EWC term is parked for its passes (bf16 mode) or written when step() runs (modes without a
            # 16-bit shadow): the returned tensor holds the penalty AFTER the step either way (REF/train/visionlanguage_tasks/train_vqa.py:160-170)
            from climb_amd.cl_algorithms import EWC
            ewc = EWC(types.SimpleNamespace(ewc_fisher_sample_percentage=0.01, ewc_loss_weight=100.0))
            P = {n: p.detach().cpu() for n, p in model.named_parameters()}
            fisher, star = _ewc_state(P, 5)
            ewc.set_task_state("nlvr2", model, fisher, star)
            opt = model.create_optimizer({"lr": 1e-4, "weight_decay": 1e-2, "adam_epsilon": 1e-8})
            opt.zero_grad()
            loss, (_, logits), task, eloss = model.fused_forward_backward("vqa", images, texts, target, ewc, optimizer=opt)
            opt.step()
            opt.zero_grad()
            out.append((float(loss) + float(eloss), logits.detach().float().cpu(), {n: p.detach().float().cpu() for n, p in model.named_parameters()}))
        else:
            opt = model.create_optimizer({"lr": 1e-3, "weight_decay": 1e-2, "adam_epsilon": 1e-8})
            opt.zero_grad()
            for _ in range(3):
                loss, (_, logits), _, _ = model.fused_forward_backward("vqa", images, texts, target, optimizer=opt)
                opt.step()
                opt.zero_grad()
            out.append((float(loss), logits.detach().float().cpu(), {n: p.detach().float().cpu() for n, p in model.named_parameters()}))
    (la, za, ga), (lb, zb, gb) = out
    # (after optimizer steps at 10 x the reference's lr the +-lr steps of near-zero-gradient elements reach the logits at the 1e-3 level in ANY pair of
    # runs, tests/test_gpu_parity.py::test_hipgraph_replay_matches_eager measured it: the forward of that path is compared at that level)
    ftol = 3e-3 if path == "optimizer_steps" else 2e-4          # (ewc_named_optimizer: the forward ran BEFORE its one step: tight)
    assert abs(la - lb) <= ftol * abs(la), (la, lb)
    assert _rel(zb, za) < ftol
    if path != "optimizer_steps":
        assert torch.equal(zb.argmax(-1), za.argmax(-1))
    if path == "ewc_named_optimizer":
        # one AdamW step of lr 1e-4 from identical moments: every element moves by ~lr sign(g); the EWC term (lam F (theta - theta*), F ~ 1e-4 x 100)
        # dominates most encoder elements identically in both modes.  Compare the update as a whole, per tensor.
        from tests.test_gpu_parity import _seeded_params
        P0 = _seeded_params(["vqa", "nlvr2"], 42)
        worst = (0.0, None)
        for n in ga:
            if n.endswith("attention.key.bias") or not n.startswith(("vilt_encoder.", "task_layer.vqa.")):
                continue
            upd = float((ga[n] - P0[n]).double().norm())
            if upd > 0:
                worst = max(worst, (float((gb[n] - ga[n]).double().norm()) / upd, n))
        print(f"{path}: worst relative difference of a tensor's update {worst[0]:.2e} ({worst[1]})")
        # (Adam's FIRST step is lr sign(g) for every element: an element whose gradient is rounding noise flips between any two runs and differs by 2 lr;
        # 0.1 % of a tensor's elements flipping = 6.5e-2 of its update's norm -- measured 6.5e-2 on the worst tensor.  A wrong or missing EWC term would
        # flip the bulk of the encoder's elements: order 1.)
        assert worst[0] < 0.2
    elif path == "optimizer_steps":
        # Adam's first steps move every element by ~lr whatever its gradient's size, so an element whose gradient is rounding noise steps +-lr in either
        # mode: compare each tensor's three-step UPDATE as a whole (the norm of the difference against the norm of the update), not element-wise
        from tests.test_gpu_parity import _seeded_params
        P0 = _seeded_params(["vqa", "nlvr2"], 42)
        worst = (0.0, None)
        for n in ga:
            if n.endswith("attention.key.bias") or not n.startswith(("vilt_encoder.", "task_layer.vqa.")):
                continue
            upd = float((ga[n] - P0[n]).double().norm())
            if upd > 0:
                worst = max(worst, (float((gb[n] - ga[n]).double().norm()) / upd, n))
        print(f"{path}: worst relative difference of a tensor's three-step update {worst[0]:.2e} ({worst[1]})")
        assert worst[0] < 6e-2          # (measured 3.0e-2 on a layer-10 query bias: 768 small gradients, a handful of them at the noise level)
    else:
        worst = _grad_err(ga, gb)
        print(f"{path}: worst per-tensor gradient difference {worst[0]:.2e} ({worst[1]})")
        assert worst[0] < 3e-4


@pytest.mark.parametrize("M,N,K", [(1536, 768, 256), (3072, 3072, 768)])
def test_gemm_split_nt_activation_epilogues(M, N, K):
    """GELU / x GELU' as epilogues of the split NT launch (epi 10: C = u = A W^T + b in fp32, out = planes of gelu(u); epi 11: out = planes of
    (A W^T) * gelu'(aux)) against float64 with the exact erf forms: the erf approximation (1.5e-7) and the planes' 2^-17 are all that separates them"""
    from climb_amd import _lib
    dev = _dev()
    assert _lib.query_arg("climb_gemm_split_nt_takes_act", M, N, K) == 1 and _lib.query_arg("climb_gemm_split_nt_takes_act", 384, N, K) == 0
    g = torch.Generator(device=dev).manual_seed(M + N)
    A = torch.randn(M, K, device=dev, generator=g)
    W = torch.randn(N, K, device=dev, generator=g) * (2.0 / math.sqrt(K))          # pre-activations of a few units: both GELU tails are visited
    bias = torch.randn(N, device=dev, generator=g)
    uaux = torch.randn(M, N, device=dev, generator=g) * 2.5
    As, Ws = _split_dev(A), _split_dev(W)
    exact = A.double().cpu() @ W.double().cpu().t()
    u = torch.full((M, N), float("nan"), device=dev)
    out = torch.empty((2, M, N), dtype=torch.bfloat16, device=dev)
    _lib.call("climb_gemm_split_nt_act", As, K, M * K, Ws, K, N * K, u, N, out, N, M * N, M, N, K, bias, 10, None, 0, _st())
    torch.cuda.synchronize()
    uref = exact + bias.double().cpu()
    assert _rel(u, uref) < 4e-5
    assert _rel(out[0].double() + out[1].double(), gelu(uref)) < 4e-5
    assert bool((out[1].float().abs() <= out[0].float().abs() * 2.0 ** -8 + 1e-30).all())          # a (hi, lo) pair: |lo| <= half an ulp of hi
    out2 = torch.empty_like(out)
    _lib.call("climb_gemm_split_nt_act", As, K, M * K, Ws, K, N * K, None, 0, out2, N, M * N, M, N, K, None, 11, uaux, N, _st())
    torch.cuda.synchronize()
    assert _rel(out2[0].double() + out2[1].double(), exact * dgelu(uaux.double().cpu())) < 4e-5
    # the erf form itself, element-wise, through a product with the identity: gelu_as(x) against the exact erf form over both tails
    lib = _lib.load()
    assert lib.climb_gemm_split_nt_act(As.data_ptr(), K, M * K, Ws.data_ptr(), K, N * K, None, 0, out.data_ptr(), N, M * N, M, N, K, None, 10, None, 0, _st()) == -1      # epi 10 needs C
    assert lib.climb_gemm_split_nt_act(As.data_ptr(), K, M * K, Ws.data_ptr(), K, N * K, None, 0, out.data_ptr(), N, M * N, 384, N, K, None, 11, uaux.data_ptr(), N, _st()) == -2


def test_bf16x3_optimizer_in_the_weight_gradient_epilogue_is_the_same_training_step(monkeypatch):
    """r06: the split mode's grouped weight-gradient launch with AdamW in its epilogue (p, m, v and the hi / lo planes of both shadows written there, the flat
    pass skipping those matrices, the planes of what the flat pass updates re-split one by one) against the same steps with the plain launch + the flat pass over
    every parameter + a full plane refresh (CLIMB_AMD_FUSED_ADAMW=0): same parameters after four steps (the update is one shared function; only the stream-K
    tiles' atomic order differs), same loss on the way -- which a stale operand plane would break from the second step on."""
    from oracle import vilt_oracle as vo
    from tests.test_gpu_parity import enc_to_inputs, make_model
    _dev()
    res = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("CLIMB_AMD_FUSED_ADAMW", fused)
        model, P = make_model(["vqa"], 42, precision="bf16x3")
        model.train()
        opt = model.create_optimizer({"lr": 1e-4, "weight_decay": 1e-2, "adam_epsilon": 1e-8})
        opt.zero_grad()
        eng = model._host.engine()
        losses, used = [], []
        for s in range(4):
            enc = vo.synthetic_encodings(2, seed=300 + s)
            images, texts = enc_to_inputs(enc)
            loss, _, _, _ = model.fused_forward_backward("vqa", images, texts, vo.synthetic_vqa_targets(2, seed=300 + s), optimizer=opt)
            used.append(bool(eng._dw_deferred))
            opt.step()
            opt.zero_grad()
            losses.append(float(loss))
            if s == 0:
                moments = (opt._m.detach().clone(), opt._v.detach().clone())
        assert all(used) == (fused == "1"), used          # the launch was held back for the optimizer exactly when asked
        # the optimizer state is sane (the first build of this epilogue stored stray register values into m / v: negative second moments, NaN two steps later)
        assert bool(torch.isfinite(eng.flat).all()) and bool(torch.isfinite(opt._m).all()) and bool(torch.isfinite(opt._v).all())
        assert bool((opt._v >= 0).all()), "negative second moment"
        # ... and the operand planes ARE the split of the fp32 parameters, in both layouts, for every GEMM weight (epilogue-written or re-split)
        eng.refresh_shadow()
        torch.cuda.synchronize()
        tot, tt = eng.layout.total, eng._shadow_t.numel() // 2
        names = [(n, N * K, N, K) for n, N, K in eng._linear_weight_names()]
        pw = next(n for n in eng.layout.offset if n.endswith("patch_embeddings.projection.weight"))
        names.append((pw, eng.layout.numel(pw), None, None))
        for n, numel, N, K in names:
            o = eng.layout.offset[n]
            w = eng.flat[o:o + numel]
            hi = w.to(torch.bfloat16)
            lo = (w - hi.float()).to(torch.bfloat16)
            assert torch.equal(eng._shadow[o:o + numel], hi) and torch.equal(eng._shadow[tot + o:tot + o + numel], lo), f"operand planes of {n}"
            if N is not None:
                t = eng._t_off[n]
                assert torch.equal(eng._shadow_t[t:t + numel].view(K, N), hi.view(N, K).t()), f"transposed hi plane of {n}"
                assert torch.equal(eng._shadow_t[tt + t:tt + t + numel].view(K, N), lo.view(N, K).t()), f"transposed lo plane of {n}"
        res[fused] = (losses, {n: p.detach().float().cpu() for n, p in model.named_parameters()}, moments)
        del model, opt
    # the moments after the FIRST step, element by element: both runs start from the same weights, so the epilogue's m / v are the flat pass's up to the atomics'
    # order in stream-K tiles (this is the comparison that found the store-data hazard: 384 elements of one matrix held stray register values, p was exact)
    for which, a, b in zip("mv", res["1"][2], res["0"][2]):
        bad = int(((a - b).abs() > 1e-6 * float(b.abs().max()) + 1e-3 * b.abs()).sum())
        assert bad == 0, f"{bad} elements of {which} differ between the epilogue and the flat pass after one step"
    la, lb = res["1"][0], res["0"][0]
    assert all(math.isfinite(x) for x in la + lb), (la, lb)
    assert max(abs(a - b) / abs(b) for a, b in zip(la, lb)) < 1e-5, (la, lb)
    # parameters: an element whose gradient sits at the summation order's noise level takes +-lr steps of either sign (Adam normalises; the zero-gradient
    # key biases and the rows of unused word embeddings are all of that kind) -- at most four lr steps apart; a weight matrix as a whole moves the same way
    diffs = {n: float((res["1"][1][n] - res["0"][1][n]).abs().max()) for n in res["0"][1]}
    worst = max((v, n) for n, v in diffs.items())
    rel = max((float((res["1"][1][n] - res["0"][1][n]).norm() / (res["0"][1][n] - P[n].float()).norm()), n) for n in diffs if res["0"][1][n].numel() >= 768 * 768 and "word_embeddings" not in n)
    print(f"fused vs flat AdamW in bf16x3 after four steps: worst element {worst[0]:.2e} ({worst[1]}); worst matrix, relative to its update {rel[0]:.2e} ({rel[1]})")
    assert worst[0] < 4 * 1e-4 * 1.3
    assert rel[0] < 0.05
