"""Probe (r03): de-phased start of the persistent NT kernel's workgroups (climb_set_option 16).  All 256 workgroups of a multi-round launch
start together and every tile costs the same, so the chip alternates between "everybody in the k-loop" (HBM idle) and "everybody storing" (matrix
pipes idle).  With the knob, workgroup group g holds back g x unit x 64 clocks once, so that one group's epilogue falls under the others' k-loops.
Scans the unit for the layer's multi-round NT products (QKV, up-projection + GELU, x GELU'), two and four groups; results must stay bit-identical.
GPU box:  python tools/probe/nt_dephase.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from climb_amd import _lib

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
t16 = _lib.torch_h16()
st = lambda: torch.cuda.current_stream().cuda_stream


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


M = 12288
settings = [0, 100, 200, 1025, 1050, 1075, 2012, 2025, 2037, 3012]
if len(sys.argv) > 1:
    settings = [int(a) for a in sys.argv[1:]]
print("option 16 value: v % 1000 = hold-back unit (x 64 clocks), v / 1000 = k: 2 << k groups.  us per launch (bit-identical to v = 0: yes / NO)")
print("shape".ljust(28) + "".join(f"{v:>10d}" for v in settings))
for name, N, K, cdt, epi in (("qkv fwd", 2304, 768, 1, 0), ("up fwd gelu", 3072, 768, 1, 1), ("du dgelu", 3072, 768, 1, 3), ("down fwd +res", 768, 3072, 0, 2), ("dxn", 768, 2304, 1, 0)):
    g = torch.Generator(device="cpu").manual_seed(1)
    A = torch.randn(M, K, generator=g).to(dev).to(t16)
    W = (torch.randn(N, K, generator=g) * 0.05).to(dev).to(t16)
    bias = torch.randn(N, generator=g).to(dev)
    aux = torch.randn(M, N, device=dev) if epi == 2 else (torch.randn(M, N, device=dev).to(t16) if epi == 3 else None)
    auxo = torch.empty(M, N, device=dev, dtype=t16) if epi == 1 else None
    ref = None
    ts = {v: [] for v in settings}
    ok = {v: True for v in settings}
    C = torch.zeros(M, N, device=dev, dtype=t16 if cdt else torch.float32)
    fn = lambda: _lib.call("climb_gemm_bf16_nt", A, K, W, K, C, N, cdt, M, N, K, bias, epi, aux, N, auxo, N, None, 0, st())
    timeit(fn, iters=100, warm=20)          # clocks settled before the first setting is timed
    for rep in range(5):                    # settings interleaved, median of 5 passes: drift hits every column alike
        for v in settings:
            _lib.call("climb_set_option", 16, v)
            C.zero_()
            ts[v].append(timeit(fn))
            torch.cuda.synchronize()
            if ref is None:
                ref = C.clone()
            ok[v] = ok[v] and bool(torch.equal(C, ref))
    _lib.call("climb_set_option", 16, 0)
    print(f"{name:14s} N={N:5d} K={K:5d}" + "".join(f"{sorted(ts[v])[2]*1e6:8.1f}{'' if ok[v] else '!'}".rjust(10) for v in settings))
