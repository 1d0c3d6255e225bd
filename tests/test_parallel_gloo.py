"""N>1 path on CPU: world_size-2 gloo processes drive climb_amd.parallel.GradientAllReducer exactly the way the engine
does (ranges reported in backward order), and check that every gradient element is averaged exactly once and that the
replicas-in-sync detector works."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from climb_amd.layout import FlatLayout, TASK_ARITH


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeEngine:
    """The slice of ViltEngine the reducer touches: flat params / grads + the layout's ranges."""

    def __init__(self, layout, rank):
        self.layout = layout
        g = torch.Generator().manual_seed(100 + rank)
        self.grad = torch.randn(layout.total, generator=g)
        self.flat = torch.zeros(layout.total)
        self.grad_ready_hook = None

    def params_updated(self, shadow_fresh=False):
        pass

    def backward_order(self, task):
        lay = self.layout
        yield lay.head_range[task]
        yield lay.top_range
        for i in range(lay.cfg["layers"] - 1, -1, -1):
            yield lay.layer_range[i]
        yield lay.embed_range


def _worker(rank, world, port, q, compress="none"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from climb_amd.parallel import GradientAllReducer
        layout = FlatLayout(["vqa", "nlvr2"], TASK_ARITH)
        eng = _FakeEngine(layout, rank)
        expect = sum(_FakeEngine(layout, r).grad for r in range(world)) / world
        S = 1024.0
        if compress == "fp16":          # the IEEE-half build: ranges reach the hook still multiplied by the engine's loss scale
            eng.h16, eng.precision, eng.loss_scale = "fp16", "bf16", S
            eng.grad.mul_(S)
        mine = eng.grad.clone()
        red = GradientAllReducer(None, compress=compress)
        red.attach(eng)
        red.begin()
        for lo, hi in eng.backward_order("vqa"):
            eng.grad_ready_hook(lo, hi)
        red.finish()
        if compress == "bf16":     # what the wire carried: each rank's gradient rounded to bf16, summed in bf16
            expect = (sum(_FakeEngine(layout, r).grad.bfloat16() for r in range(world))).float() / world
        if compress == "fp16":     # scaled, rounded to half, summed in half; the reducer divides the scale out with the average
            assert red.takes_scaled
            expect = (sum((_FakeEngine(layout, r).grad * S).half() for r in range(world))).float() / (world * S)
        lo, hi = layout.head_range["nlvr2"]          # the head that got no gradient is not touched by any collective
        ok_untouched = torch.equal(eng.grad[lo:hi], mine[lo:hi])
        mask = torch.ones(layout.total, dtype=torch.bool)
        mask[lo:hi] = False
        ok_avg = torch.allclose(eng.grad[mask], expect[mask], rtol=0, atol=1e-6 if compress == "none" else 2e-2)
        ok_avg = ok_avg and float((eng.grad[mask] - expect[mask]).norm() / expect[mask].norm()) < (1e-6 if compress == "none" else 4e-3)
        ncoll = red.bytes_reduced
        # replicas-in-sync detector
        eng.flat.fill_(1.0)
        in_sync = red.replicas_in_sync()
        if rank == 1:
            eng.flat[123] += 1e-3
        out_of_sync = not red.replicas_in_sync()
        q.put((rank, ok_untouched, ok_avg, in_sync, out_of_sync, ncoll, layout.total - (hi - lo)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("compress", ["none", "bf16", "fp16"])
def test_bucketed_allreduce_world2(compress):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, compress)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_untouched, ok_avg, in_sync, out_of_sync, nbytes, nelem in res:
        assert ok_untouched and ok_avg, (rank, ok_untouched, ok_avg)
        assert in_sync and out_of_sync
        assert nbytes == (4 if compress == "none" else 2) * nelem, "every reported gradient element is reduced exactly once"


def _reserve_worker(rank, world, port, q, overlap):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from climb_amd.parallel import GradientAllReducer
        layout = FlatLayout(["vqa"], TASK_ARITH)
        eng = _FakeEngine(layout, rank)
        calls = []
        eng.set_cu_reserve = calls.append
        red = GradientAllReducer(None, compress="none", overlap=overlap)
        red.reserve_cus = 32
        red.attach(eng)
        seen_during = []
        for _ in range(2):          # two steps: the reserve is taken and given back every step
            red.begin()
            for lo, hi in eng.backward_order("vqa"):
                eng.grad_ready_hook(lo, hi)
                seen_during.append(list(calls))
            red.finish()
        q.put((rank, calls, seen_during[0], red.bytes_reduced))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
def test_cu_reserve_is_held_only_while_collectives_run_under_the_backward(overlap):
    """GradientAllReducer.reserve_cus (bench.py tries it in its warm-up): from the first collective launched UNDER the backward until finish() the
    engine is told to leave CUs to RCCL (persistent GEMM grids shrink); with deferred collectives nothing runs next to the backward and the engine is
    never asked.  The averaged gradients are the same either way (checked by the tests above: the reserve only sizes launches)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reserve_worker, args=(r, world, port, q, overlap)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, calls, first_hook_calls, nbytes in res:
        assert nbytes > 0
        if overlap:
            assert calls == [32, 0, 32, 0], calls          # taken at the first collective of each step, given back by finish()
            assert first_hook_calls in ([], [32])          # (the head's range may be too small for a collective of its own)
        else:
            assert calls == [], calls


def test_payload_type_follows_the_librarys_16_bit_type():
    """On the IEEE-half build of the library the 16-bit cast kernels produce half, whose range holds the gradients only while they carry the
    loss scale: the reducer's 16-bit payload is "fp16" there (scaled ranges, scale divided out in finish(); opt-in since r05) and an explicit bf16 one
    is refused; on the bf16 build the throughput mode defaults to bf16."""
    from climb_amd.layout import FlatLayout, TASK_ARITH
    from climb_amd.parallel import GradientAllReducer
    lay = FlatLayout(["vqa"], TASK_ARITH)
    # (r05: the half build DEFAULTS to the fp32 payload -- a scaled weight gradient can leave half's range when ranks are summed -- "fp16" is opt-in)
    for h16, precision, want in (("bf16", "bf16", "bf16"), ("fp16", "bf16", "none"), (None, "fp32", "none")):
        eng = _FakeEngine(lay, 0)
        eng.h16, eng.precision = h16, precision
        r = GradientAllReducer()
        r.attach(eng)
        assert r.compress == want, (h16, r.compress)
    eng = _FakeEngine(lay, 0)
    eng.h16, eng.precision = "fp16", "bf16"
    with pytest.raises(ValueError, match="IEEE half"):
        GradientAllReducer(compress="bf16").attach(eng)
    r = GradientAllReducer(compress="fp16")
    r.attach(eng)
    assert not r.takes_scaled          # one rank: nothing is cast, the engine unscales as usual
    eng.h16 = "bf16"
    with pytest.raises(ValueError, match="needs the fp16 build"):
        GradientAllReducer(compress="fp16").attach(eng)


def test_bucket_merging_single_process():
    """Small adjacent ranges (final norm + pooler) ride with the neighbouring layer instead of paying their own collective."""
    from climb_amd.parallel import GradientAllReducer
    layout = FlatLayout(["vqa"], TASK_ARITH)
    red = GradientAllReducer(None)
    launched = []
    red._launch = lambda lo, hi: launched.append((lo, hi))
    eng = _FakeEngine(layout, 0)
    red.attach(eng)
    red.begin()
    for lo, hi in eng.backward_order("vqa"):
        red.on_ready(lo, hi)
    assert not red._small, "nothing small is left waiting: the final norm + pooler rode with layer 11"
    assert sum(hi - lo for lo, hi in launched) == layout.total
    assert len(launched) == 1 + 12 + 1                       # head | layer 11 (+ final norm/pooler) .. layer 0 | embeddings
    assert all(hi - lo >= red.min_bucket for lo, hi in launched)
    srt = sorted(launched)
    assert all(a[1] == b[0] for a, b in zip(srt, srt[1:])), "buckets tile the buffer without gaps or overlap"


def _payload_worker(rank, world, port, q, scenario, compress, overlap):
    """Frozen-prefix / adapter-only steps: only trainable sub-ranges are reported, fragments travel packed."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from climb_amd.engine import ViltEngine
        from climb_amd.parallel import GradientAllReducer
        adapters = {"vqa": 48} if scenario == "adapter" else {}
        layout = FlatLayout(["vqa"], TASK_ARITH, adapters=adapters)
        eng = ViltEngine(layout, torch.device("cpu"), "fp32")           # host bookkeeping only: no kernels are launched
        g = torch.Generator().manual_seed(100 + rank)
        eng.grad = torch.randn(layout.total, generator=g)
        eng.flat = torch.zeros(layout.total)
        mine = eng.grad.clone()
        if scenario == "adapter":           # base frozen, adapter + head trainable
            first = 0
            for n in layout.shapes:
                eng.requires_grad[n] = (".adapters.vqa." in n) or n.startswith("task_layer.")
        else:                               # bottom 9 layers + embeddings frozen
            first = 9
            for n in layout.shapes:
                eng.requires_grad[n] = not (n.startswith("vilt_encoder.vilt.embeddings") or any(f".layer.{i}." in n for i in range(9)))
        red = GradientAllReducer(None, compress=compress, overlap=overlap)
        red.attach(eng)
        red.begin()
        eng._ready(*layout.head_range["vqa"])                          # the order encoder_backward reports ranges in
        eng._ready(*layout.top_range)
        for i in range(layout.cfg["layers"] - 1, first - 1, -1):
            eng._ready(*layout.layer_range[i])
        red.finish()
        trainable = torch.zeros(layout.total, dtype=torch.bool)
        for n, start, length in layout.segments():
            if eng.requires_grad[n] and (scenario == "adapter" or True):
                trainable[start:start + length] = True
        grads = [torch.randn(layout.total, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
        expect = (sum(x.bfloat16() for x in grads).float() if compress == "bf16" else sum(grads)) / world
        tol = 2e-2 if compress == "bf16" else 1e-6
        ok_avg = torch.allclose(eng.grad[trainable], expect[trainable], rtol=0, atol=tol)
        ok_frozen = torch.equal(eng.grad[~trainable], mine[~trainable])
        q.put((rank, ok_avg, ok_frozen, red.bytes_reduced, int(trainable.sum()), red.collectives))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("scenario,compress,overlap", [("adapter", "none", True), ("adapter", "bf16", False), ("bottom9", "bf16", True), ("bottom9", "none", False)])
def test_payload_shrinks_to_trainable_ranges_world2(scenario, compress, overlap):
    """with a frozen base (adapters) or a frozen prefix (freeze_bottom_k_layers) only the trainable ranges cross the
    wire -- 1.8 M adapter + 6 M head parameters instead of 120 M -- and the 24 adapter fragments of a step travel as a handful of
    packed collectives, with and without overlap."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_payload_worker, args=(r, world, port, q, scenario, compress, overlap)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_avg, ok_frozen, nbytes, ntrain, ncoll in res:
        assert ok_avg and ok_frozen, (rank, ok_avg, ok_frozen)
        assert nbytes == (2 if compress == "bf16" else 4) * ntrain, "exactly the trainable elements are reduced, once"
        if scenario == "adapter":
            assert ntrain < 9_000_000 and ncoll <= 3, (ntrain, ncoll)          # head in place + the adapter fragments packed
        else:
            assert 25_000_000 < ntrain < 40_000_000 and ncoll <= 5, (ntrain, ncoll)
