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
        opt(17, 3)
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


def bench(M=12288, passes=5, rotate=1, settings=((17, 0, 18, 0), (17, 3, 18, 0)), cold=False, only=None, zeros=False):
    """rotate > 1: every launch writes a different output set (rotate x the outputs > the 256 MB Infinity Cache: stores really go to HBM, as in a step).
    cold: as a step has it -- the A operand was WRITTEN by the launch before (a copy into one of 12 buffers; its time, measured alone, is subtracted) and
    the weights are one of 12 copies that were last read 11 launches ago"""
    shapes = [("qkv fwd", 2304, 768, 1, 0), ("out fwd +res", 768, 768, 0, 2), ("up fwd gelu", 3072, 768, 1, 1), ("down fwd +res", 768, 3072, 0, 2),
              ("du dgelu", 3072, 768, 1, 3), ("dhn", 768, 3072, 1, 0), ("dctx", 768, 768, 1, 0), ("dxn", 768, 2304, 1, 0)]
    tot = {i: 0.0 for i in range(len(settings))}
    flops = 0.0
    for name, N, K, cdt, epi in shapes:
        if only and name not in only:
            continue
        A = torch.randn(M, K, device=dev).to(H16)
        W = (torch.randn(N, K, device=dev) * 0.05).to(H16)
        if zeros:
            A.zero_(); W.zero_()
        nrot = 12 if cold else 1
        As = [torch.empty_like(A) for _ in range(nrot)] if cold else [A]
        Ws = [W.clone() for _ in range(nrot)]
        Cs = [torch.empty(M, N, device=dev, dtype=H16 if cdt else torch.float32) for _ in range(rotate)]
        bias = torch.randn(N, device=dev)
        auxs = [torch.randn(M, N, device=dev) if epi == 2 else (torch.randn(M, N, device=dev).to(H16) if epi == 3 else None) for _ in range(rotate)]
        auxos = [torch.empty(M, N, device=dev, dtype=H16) if epi == 1 else None for _ in range(rotate)]
        ctr = [0]

        def fn():
            i = ctr[0] % rotate
            j = ctr[0] % nrot
            ctr[0] += 1
            if cold:
                As[j].copy_(A)
            _lib.call("climb_gemm_bf16_nt", As[j], K, Ws[j], K, Cs[i], N, cdt, M, N, K, bias, epi, auxs[i], N, auxos[i], N, None, 0, st())
        t_copy = 0.0
        if cold:
            cc = [0]

            def fc():
                cc[0] += 1
                As[cc[0] % nrot].copy_(A)
            t_copy = min(timeit(fc) for _ in range(3))
        ts = {i: [] for i in range(len(settings))}
        for _ in range(passes):
            for i, sv in enumerate(settings):
                for k in range(0, len(sv), 2):
                    opt(sv[k], sv[k + 1])
                ts[i].append(timeit(fn) - t_copy)
        med = {v: sorted(ts[v])[len(ts[v]) // 2] for v in ts}
        f = 2.0 * M * N * K
        flops += f
        for v in med:
            tot[v] += med[v]
        print(f"NT {name:14s} N={N:5d} K={K:5d}: " + " | ".join(f"{settings[v]} {med[v]:7.1f} us {f/med[v]/1e6:7.1f} TF" for v in med), flush=True)
    print("NT total per layer: " + " | ".join(f"{settings[v]} {tot[v]:.1f} us ({flops/tot[v]/1e6:.1f} TF)" for v in tot))
    opt(17, 1)
    opt(18, 0)




def seq(M=12288, iters=12, rotate=6):
    """run under `rocprofv3 --kernel-trace --stats`: every GEMM reads an A operand that the launch before it has just WRITTEN (a copy kernel, like the
    LayerNorm in front of the forward GEMMs of a step) and writes outputs nothing has touched for a while -- the in-step situation, kernel by kernel.
    Variants (the kernel names in the trace are the same; the variants run in this order, `iters` launches each, per shape and kernel):
      0: one A buffer, one W;  1: 12 A buffers (each written just before its GEMM);  2: 12 A buffers and 12 W copies (cold weights)"""
    shapes = [("qkv fwd", 2304, 768, 1, 0), ("up fwd gelu", 3072, 768, 1, 1)]
    for name, N, K, cdt, epi in shapes:
        src = torch.randn(M, K, device=dev).to(H16)
        As = [torch.empty_like(src) for _ in range(12)]
        Ws = [(torch.randn(N, K, device=dev) * 0.05).to(H16) for _ in range(12)]
        junk = torch.empty(64 << 20, device=dev)          # 256 MB: pushes the weights out of the Infinity Cache between uses
        Cs = [torch.empty(M, N, device=dev, dtype=H16 if cdt else torch.float32) for _ in range(rotate)]
        bias = torch.randn(N, device=dev)
        auxos = [torch.empty(M, N, device=dev, dtype=H16) if epi == 1 else None for _ in range(rotate)]
        for variant in (0, 1, 2):
            for v in (0, 3):
                opt(17, v)
                for it in range(iters):
                    i = it % rotate
                    A = As[it % 12 if variant >= 1 else 0]
                    W = Ws[it % 12 if variant >= 2 else 0]
                    if variant >= 2 and it % 12 == 0:
                        junk.zero_()
                    A.copy_(src)
                    _lib.call("climb_gemm_bf16_nt", A, K, W, K, Cs[i], N, cdt, M, N, K, bias, epi, None, N, auxos[i], N, None, 0, st())
                torch.cuda.synchronize()
    opt(17, 1)


if __name__ == "__main__" and "--seq" in sys.argv:
    seq()
    sys.exit(0)
if __name__ == "__main__":
    quick = "--quick" in sys.argv
    ok = True
    for M, N, K in [(1536, 192, 640), (1536, 384, 704), (3072, 768, 768), (12288, 768, 768)] + ([] if quick else [(12288, 2304, 768), (12288, 3072, 768), (12288, 768, 3072)]):
        ok = check(M, N, K) and ok
    if "--no-bench" not in sys.argv:
        bench(settings=((17, 0), (17, 3), (17, 2)))
    if "--rotate" in sys.argv:
        print("== outputs rotated over 6 sets (HBM-resident, as in a step)")
        bench(rotate=6, passes=3, settings=((17, 0), (17, 3), (17, 2)))
    if "--anatomy" in sys.argv:
        print("== measurement builds of the 16-bit NONE kernel (option 18 = 100 + bits; 1: no DMA in the windows, 2: no epilogue in the windows, 4: all DMA pieces in slots 4..15)")
        for cold in (False, True):
            bench(rotate=6, passes=3, settings=((17, 0, 18, 0), (17, 3, 18, 0), (17, 3, 18, 101), (17, 3, 18, 102), (17, 3, 18, 103), (17, 3, 18, 104)), cold=cold,
                  only=("qkv fwd", "dhn", "dctx"))
    if "--anatomy2" in sys.argv:
        print("== r05: the bare window (103 = no DMA, no epilogue) minus its barrier (111), minus its fragment reads (119), minus both (127: MFMAs only)")
        bench(rotate=6, passes=3, settings=((17, 3, 18, 0), (17, 3, 18, 103), (17, 3, 18, 111), (17, 3, 18, 119), (17, 3, 18, 127)), only=("qkv fwd", "dhn"))
    if "--zeros" in sys.argv:
        print("== zero operands (the matrix pipes toggle nothing: what the clock does when the power budget is not the limit)")
        bench(rotate=6, passes=3, settings=((17, 0), (17, 3)), zeros=True)
    if "--cold" in sys.argv:
        print("== as in a step: outputs rotated over 6 sets, A written by the launch before, weights not touched for 11 launches")
        bench(rotate=6, passes=3, settings=((17, 0, 18, 0), (17, 3, 18, 0), (17, 2, 18, 0), (17, 3, 18, 2)), cold=True)
    sys.exit(0 if ok else 1)

