"""PrefetchLoader (row F1, host logic): order, bounded look-ahead, error delivery, early exit; on the GPU the side-stream hand-off."""
import threading
import time

import pytest
import torch

from climb_amd.data import PrefetchLoader


def test_order_and_bounded_lookahead():
    produced, lock = [], threading.Lock()

    def prepare(i):
        with lock:
            produced.append(i)
        return i * i

    seen = []
    for v in PrefetchLoader(range(20), prepare, depth=2):
        time.sleep(0.01)                       # a slow consumer: the worker may run ahead by `depth` (+ the one it holds)
        with lock:
            ahead = len(produced) - len(seen)
        assert ahead <= 2 + 2, ahead
        seen.append(v)
    assert seen == [i * i for i in range(20)]


def test_worker_exception_surfaces_at_its_batch():
    def prepare(i):
        if i == 3:
            raise ValueError("bad batch 3")
        return i

    got = []
    with pytest.raises(ValueError, match="bad batch 3"):
        for v in PrefetchLoader(range(10), prepare, depth=3):
            got.append(v)
    assert got == [0, 1, 2]


def test_early_break_stops_the_worker():
    calls = []
    it = PrefetchLoader(range(10 ** 6), lambda i: calls.append(i) or i, depth=2)
    for v in it:
        if v == 5:
            break
    time.sleep(0.3)
    n = len(calls)
    time.sleep(0.3)
    assert len(calls) == n and n < 50           # the worker is not still consuming the source
    with pytest.raises(ValueError):
        PrefetchLoader([], lambda x: x, depth=0)


@pytest.mark.gpu
def test_side_stream_handoff_is_ordered():
    """Work enqueued by the worker on its side stream is visible to kernels the consumer enqueues afterwards on its own stream."""
    dev = torch.device("cuda:0")

    def prepare(i):
        x = torch.full((1 << 22,), float(i), device=dev)
        for _ in range(20):
            x = x * 1.0 + 0.0                  # keep the side stream busy so a missing dependency would be observable
        return x

    for i, x in enumerate(PrefetchLoader(range(8), prepare, depth=2, device=dev)):
        assert float(x.sum()) == float(i) * (1 << 22)
