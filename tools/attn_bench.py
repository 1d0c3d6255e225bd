"""Micro-benchmark of the bf16 attention kernels at the benchmark shape (B=64, 12 heads, S_pad=192)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from climb_amd import _lib

dev = torch.device("cuda:0")
B, heads, d, S_pad = 64, 12, 64, int(os.environ.get("S_PAD", 192))
H = heads * d
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


qkv = torch.randn(B * S_pad, 3 * H, device=dev).bfloat16()
bias = torch.zeros(B, S_pad, device=dev)
bias[:, 185:] = -3e38
ctx = torch.empty(B * S_pad, H, device=dev, dtype=torch.bfloat16)
dctx = torch.randn(B * S_pad, H, device=dev).bfloat16()
lse = torch.empty(B, heads, S_pad, device=dev)
delta = torch.empty(B, heads, S_pad, device=dev)
dqkv = torch.empty(B * S_pad, 3 * H, device=dev, dtype=torch.bfloat16)
f = 4.0 * S_pad * S_pad * d * heads * B
t = timeit(lambda: _lib.call("climb_attn_fwd_bf16", qkv, bias, ctx, lse, B, S_pad, heads, d, st()))
print(f"fwd  {t*1e6:7.1f} us  {f/t/1e12:6.1f} TF")
t = timeit(lambda: _lib.call("climb_attn_delta", dctx, ctx, 1, delta, B, S_pad, heads, st()))
print(f"delta {t*1e6:6.1f} us")
outs = {}
for fused in (4, 3, 1, 0):    # r04: single pass, persistent / one workgroup per item (default up to S_pad = 128) / r03: both phases in one launch (default above) / one launch per phase
    _lib.call("climb_set_option", 13, fused)
    t = timeit(lambda: _lib.call("climb_attn_bwd_bf16", qkv, bias, dctx, ctx, lse, delta, dqkv, B, S_pad, heads, d, st()))
    outs[fused] = dqkv.float().clone()
    nprod = 5 if (fused >= 3 and S_pad <= 192) else 7
    print(f"bwd  {t*1e6:7.1f} us  {nprod / 2 * f/t/1e12:6.1f} TF ({nprod} products)  {('two launches', 'one launch, two phases', '', 'single pass', 'single pass, persistent')[fused]}")
_lib.call("climb_set_option", 13, 2)
d21 = (outs[3] - outs[1]).norm() / outs[1].norm()
print(f"single pass vs two-phase: relative L2 difference {d21:.2e} (different summation order of the 16-bit products); persistent == plain single pass: {bool(torch.equal(outs[4], outs[3]))}")
