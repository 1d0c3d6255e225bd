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
        mine = eng.grad.clone()
        red = GradientAllReducer(None, compress=compress)
        red.attach(eng)
        red.begin()
        for lo, hi in eng.backward_order("vqa"):
            eng.grad_ready_hook(lo, hi)
        red.finish()
        if compress == "bf16":     # what the wire carried: each rank's gradient rounded to bf16, summed in bf16
            expect = (sum(_FakeEngine(layout, r).grad.bfloat16() for r in range(world))).float() / world
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


@pytest.mark.parametrize("compress", ["none", "bf16"])
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
    if red._pending:
        launched.append(red._pending)
    assert sum(hi - lo for lo, hi in launched) == layout.total
    assert len(launched) == 1 + 12 + 1                       # head | layer 11 (+ final norm/pooler) .. layer 0 | embeddings
    assert all(hi - lo >= red.min_bucket for lo, hi in launched)
    srt = sorted(launched)
    assert all(a[1] == b[0] for a, b in zip(srt, srt[1:])), "buckets tile the buffer without gaps or overlap"
