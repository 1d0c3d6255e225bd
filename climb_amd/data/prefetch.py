"""Batch prefetching for the input pipeline: tokenisation, raw-image staging, the host->device copies and
the device pre-processing kernels of batch i+1 run on a worker thread and a side HIP stream while the training thread is inside
step i.  The reference does all of it on the training thread, every step (REF/modeling/vilt.py:83-96)."""
from __future__ import annotations

import queue
import threading
from typing import Callable, Iterable, Iterator, Optional

import torch

_END = object()


def _record_stream(obj, stream):
    """Tensors allocated under the worker's side stream are consumed on the training stream: tell the caching allocator, or it may
    hand their memory to the next prefetch while step kernels still read it."""
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _record_stream(v, stream)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _record_stream(v, stream)


class PrefetchLoader:
    """Wraps an iterable of raw batches.  `prepare(batch)` runs on the worker (under a side stream when `device` is a HIP device)
    and returns whatever the step consumes (e.g. the batch with `process_inputs` already applied).  At most `depth` prepared
    batches exist at a time; exceptions raised by the worker are re-raised in the consumer at the batch they belong to."""

    def __init__(self, batches: Iterable, prepare: Callable, depth: int = 2, device: Optional[torch.device] = None):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.batches, self.prepare, self.depth = batches, prepare, depth
        self.device = torch.device(device) if device is not None else None
        self._cuda = self.device is not None and self.device.type == "cuda"

    def __iter__(self) -> Iterator:
        q: "queue.Queue" = queue.Queue(maxsize=self.depth)
        stop = threading.Event()
        side = torch.cuda.Stream(device=self.device) if self._cuda else None

        def put(item) -> bool:
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    continue
            return False

        def work():
            try:
                for raw in self.batches:
                    if stop.is_set():
                        return
                    if side is not None:
                        with torch.cuda.stream(side):
                            out = self.prepare(raw)
                            ev = torch.cuda.Event()
                            ev.record(side)
                    else:
                        out, ev = self.prepare(raw), None
                    if not put((out, ev, None)):
                        return
            except BaseException as e:      # noqa: BLE001  (delivered to the consumer)
                put((None, None, e))
                return
            put(_END)

        t = threading.Thread(target=work, name="climb-prefetch", daemon=True)
        t.start()
        try:
            while True:
                item = q.get()
                if item is _END:
                    return
                out, ev, err = item
                if err is not None:
                    raise err
                if ev is not None:
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(ev)                                          # device-side dependency, no host sync
                    _record_stream(out, cur)
                yield out
        finally:
            stop.set()
            t.join(timeout=5)
