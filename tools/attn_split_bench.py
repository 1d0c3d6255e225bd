"""split-operand attention at the step's shape (B = 64, S_pad = 192, 12 heads): forward / backward launch times for LDS chunk and wave settings.
python tools/attn_split_bench.py     (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from climb_amd import _lib
dev = torch.device("cuda:0")
B, S, nh, d = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 192, 12, 64
H, M = nh * d, B * S
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(M, 3 * H, device=dev, generator=g)
kb = torch.zeros(B, S, device=dev); kb[:, S - 7:] = -3e38
dctx = torch.randn(M, H, device=dev, generator=g)
ctx = torch.empty(M, H, device=dev); ctx_s = torch.empty(2, M, H, dtype=torch.bfloat16, device=dev)
lse = torch.empty(B, nh, S, device=dev); delta = torch.empty(B, nh, S, device=dev)
dq = torch.empty(2, M, 3 * H, dtype=torch.bfloat16, device=dev)
st = torch.cuda.current_stream().cuda_stream
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
fwd = lambda: _lib.call("climb_attn_fwd_split", qkv, kb, ctx, ctx_s, M * H, lse, B, S, nh, d, st)
bwd = lambda: _lib.call("climb_attn_bwd_split", qkv, kb, dctx, lse, delta, None, dq, M * 3 * H, B, S, nh, d, st)
fwd(); _lib.call("climb_attn_delta", dctx, ctx, 0, delta, B, S, nh, st)
ref = None
for mk in (192, 96, 64):
    for nwf, nwb in ((0, 0), (4, 4), (6, 4), (8, 4), (8, 8)):
        for k, v in ((23, mk), (24, nwf), (25, nwb)):
            _lib.call("climb_set_option", k, v)
        f, b_ = t(fwd), t(bwd)
        torch.cuda.synchronize()
        snap = (ctx.clone(), dq.clone())
        if ref is None: ref = snap
        same = float((snap[0] - ref[0]).abs().max()), float((snap[1].float() - ref[1].float()).abs().max())
        print(f"S_pad {S} rows/chunk {mk:3d} waves fwd {nwf or 'auto'} bwd {nwb or 'auto'}: fwd {f:6.1f} us  bwd {b_:6.1f} us   (max diff vs first setting: ctx {same[0]:.1e}, dqkv {same[1]:.1e})")
