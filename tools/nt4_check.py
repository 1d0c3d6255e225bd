"""r04: the 4-wave / two-accumulator-set NT kernel (gemm_bf16_nt4.hip) against the kernels it replaces -- bit for bit -- and timed next to them on the
layer's eight NT shapes, settings interleaved.  Run on the GPU box:  python tools/nt4_check.py [--quick]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from climb_amd import _lib

dev = torch.device("cuda:0")
st = lambda: torch.cuda.current_stream().cuda_stream
H16 = _lib.torch_h16()


def opt(k, v):
    _lib.call("climb_set_option", k, v)


def run(A, W, bias, cdt, epi, aux, M, N, K):
    C = torch.full((M, N), float("nan"), device=dev, dtype=H16 if cdt else torch.float32)
    U = torch.full((M, N), float("nan"), device=dev, dtype=H16) if epi == 1 else None
    _lib.call("climb_gemm_bf16_nt", A, K, W, K, C, N, cdt, M, N, K, bias, epi, aux, N, U, N, None, 0, st())
    return C, U


def check(M, N, K, reps=3):
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    A = torch.randn(M, K, device=dev, generator=g).to(H16)
    W = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(H16)
    bias = torch.randn(N, device=dev, generator=g)
    R = torch.randn(M, N, device=dev, generator=g)
    Uin = torch.randn(M, N, device=dev, generator=g).to(H16)
    bad_total = 0
    for name, cdt, epi, aux, b in [("f32 none", 0, 0, None, bias), ("h16 none", 1, 0, None, bias), ("h16 none nobias", 1, 0, None, None), ("h16 gelu", 1, 1, None, bias),
                                   ("f32 resid", 0, 2, R, bias), ("h16 dgelu", 1, 3, Uin, None)]:
        opt(17, 0)
        ref = run(A, W, b, cdt, epi, aux, M, N, K)
        opt(17, 1)
        for it in range(reps):
            got = run(A, W, b, cdt, epi, aux, M, N, K)
            torch.cuda.synchronize()
            for r_, g_ in zip(ref, got):
                if r_ is None:
                    continue
                nan = int(torch.isnan(g_.float()).sum())
                bad = int((g_.float() != r_.float()).sum()) - int((torch.isnan(g_.float()) & torch.isnan(r_.float())).sum())
                if bad or nan:
                    bad_total += 1
                    d = (g_.float() - r_.float()).abs()
                    rows = (g_.float() != r_.float()).any(dim=1).nonzero().flatten()
                    cols = (g_.float() != r_.float()).any(dim=0).nonzero().flatten()
                    print(f"  MISMATCH M={M} N={N} K={K} {name} it={it}: {bad} differ, {nan} nan, max |d| {d[~torch.isnan(d)].max().item() if (~torch.isnan(d)).any() else float('nan'):.3e}; "
                          f"rows {rows[:6].tolist()}..{rows[-3:].tolist()} ({len(rows)}), cols {cols[:6].tolist()}..{cols[-3:].tolist()} ({len(cols)})")
    print(f"check M={M} N={N} K={K}: {'OK' if bad_total == 0 else 'FAILED'}", flush=True)
    return bad_total == 0


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def bench(M=12288, passes=5):
    shapes = [("qkv fwd", 2304, 768, 1, 0), ("out fwd +res", 768, 768, 0, 2), ("up fwd gelu", 3072, 768, 1, 1), ("down fwd +res", 768, 3072, 0, 2),
              ("du dgelu", 3072, 768, 1, 3), ("dhn", 768, 3072, 1, 0), ("dctx", 768, 768, 1, 0), ("dxn", 768, 2304, 1, 0)]
    tot = {0: 0.0, 1: 0.0}
    flops = 0.0
    for name, N, K, cdt, epi in shapes:
        A = torch.randn(M, K, device=dev).to(H16)
        W = (torch.randn(N, K, device=dev) * 0.05).to(H16)
        C = torch.empty(M, N, device=dev, dtype=H16 if cdt else torch.float32)
        bias = torch.randn(N, device=dev)
        aux = torch.randn(M, N, device=dev) if epi == 2 else (torch.randn(M, N, device=dev).to(H16) if epi == 3 else None)
        auxo = torch.empty(M, N, device=dev, dtype=H16) if epi == 1 else None
        fn = lambda: _lib.call("climb_gemm_bf16_nt", A, K, W, K, C, N, cdt, M, N, K, bias, epi, aux, N, auxo, N, None, 0, st())
        ts = {0: [], 1: []}
        for _ in range(passes):
            for v in (0, 1):
                opt(17, v)
                ts[v].append(timeit(fn))
        med = {v: sorted(ts[v])[len(ts[v]) // 2] for v in ts}
        f = 2.0 * M * N * K
        flops += f
        for v in med:
            tot[v] += med[v]
        print(f"NT {name:14s} N={N:5d} K={K:5d}: 8-wave {med[0]:7.1f} us {f/med[0]/1e6:7.1f} TF | nt4 {med[1]:7.1f} us {f/med[1]/1e6:7.1f} TF", flush=True)
    print(f"NT total per layer: 8-wave {tot[0]:.1f} us ({flops/tot[0]/1e6:.1f} TF) | nt4 {tot[1]:.1f} us ({flops/tot[1]/1e6:.1f} TF)")


if __name__ == "__main__":
    quick = "--quick" in sys.argv
    ok = True
    for M, N, K in [(1536, 192, 640), (1536, 384, 704), (3072, 768, 768), (12288, 768, 768)] + ([] if quick else [(12288, 2304, 768), (12288, 3072, 768), (12288, 768, 3072)]):
        ok = check(M, N, K) and ok
    if "--no-bench" not in sys.argv:
        bench()
    sys.exit(0 if ok else 1)
