"""Data parallelism for the ViLT step: one process per GPU, full replica, gradient all-reduce over RCCL/xGMI overlapped
with the backward.  The reference has no distributed code at all; this is new design, not a counterpart.

Why it is shaped like this on MI355X
  * xGMI is point-to-point (7 links x ~153 GB/s per GPU), so collectives are per-link bound: few LARGE messages beat
    many small ones.  The flat gradient buffer is laid out in forward order, so every encoder layer is ONE contiguous
    28 MB fp32 range: the engine reports `(lo, hi)` the moment a layer's weight gradients are final and that range is
    all-reduced in place -- no bucket copy-in/copy-out, no per-tensor calls.
  * The collective is enqueued with `async_op=True`: torch's RCCL process group runs it on its own HIP stream after an
    event on the compute stream, so layer i's all-reduce rides under layer i-1's backward GEMMs.
  * `finish()` makes the compute stream wait for every outstanding collective; only then are the EWC penalty gradient
    (identical on every rank, so it must NOT be summed) and the fused AdamW enqueued.
  * ReduceOp.AVG on RCCL; SUM followed by a scale on backends without AVG (gloo, used by the CPU tests).
  * Payload: fp32 (bit-faithful averaging, the parity default) or bf16 (`compress="bf16"`, the default of the bf16 throughput
    mode; SURVEY.md §8(e) "prefer bf16 payload"): the range is cast into a persistent bf16 staging buffer laid out like the
    gradient buffer, reduced there, and cast back in `finish()`.  Halves the bytes every xGMI link carries (480 -> 240 MB per
    step) for two extra streaming passes over the gradients; the rounding (2^-9 relative per element, once) is below the bf16
    GEMM noise already in those gradients.  `CLIMB_AMD_DP_COMPRESS=none|bf16` overrides.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


class GradientAllReducer:
    def __init__(self, model=None, process_group=None, min_bucket_elems: int = 4 * 1024 * 1024, broadcast: bool = True,
                 compress: Optional[str] = None):
        self.pg = process_group
        self.compress = compress            # None = decide at attach() from the engine's precision
        self._stage = None
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.min_bucket = min_bucket_elems
        self.eng = None
        self._works: List[Tuple[object, int, int]] = []
        self._pending: Optional[Tuple[int, int]] = None
        self._avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        self.bytes_reduced = 0
        if model is not None:
            host = model._host
            host.ddp = self
            self.attach(host.engine())
            if broadcast and self.world > 1:
                dist.broadcast(self.eng.flat, src=0, group=self.pg)
                self.eng.params_updated(shadow_fresh=False)

    def attach(self, eng):
        """(Re)bind to an engine: its `grad_ready_hook` fires with a flat range whose gradients are final."""
        self.eng = eng
        eng.grad_ready_hook = self.on_ready
        mode = os.environ.get("CLIMB_AMD_DP_COMPRESS") or self.compress
        if mode is None:
            mode = "bf16" if getattr(eng, "precision", "fp32") == "bf16" else "none"
        if mode not in ("none", "bf16"):
            raise ValueError(f"unknown gradient compression {mode!r}")
        self.compress = mode
        self._stage = None

    def begin(self):
        self._works.clear()
        self._pending = None

    def _launch(self, lo: int, hi: int):
        if self.world == 1:
            return
        chunk = self.eng.grad[lo:hi]
        if self.compress == "bf16":
            if self._stage is None:
                self._stage = torch.empty(self.eng.grad.numel(), dtype=torch.bfloat16, device=self.eng.grad.device)
            stage = self._stage[lo:hi]
            stage.copy_(chunk)                 # fp32 -> bf16 (round to nearest even) on the compute stream
            chunk = stage
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        work = dist.all_reduce(chunk, op=op, group=self.pg, async_op=True)
        self._works.append((work, lo, hi))
        self.bytes_reduced += (hi - lo) * chunk.element_size()

    def on_ready(self, lo: int, hi: int):
        """Ranges arrive in backward order (head, final norm + pooler, layer 11 ... layer 0, embeddings).  Adjacent small
        ranges are merged until a bucket is worth a collective."""
        if self._pending is not None:
            plo, phi = self._pending
            if hi == plo:                      # contiguous with the pending range (walking down the buffer)
                lo, hi = lo, phi
            elif lo == phi:
                lo, hi = plo, hi
            else:
                self._launch(plo, phi)
            self._pending = None
        if hi - lo >= self.min_bucket:
            self._launch(lo, hi)
        else:
            self._pending = (lo, hi)

    def finish(self):
        """Block the compute stream on every outstanding collective (no host sync on RCCL)."""
        if self._pending is not None:
            self._launch(*self._pending)
            self._pending = None
        for work, lo, hi in self._works:
            work.wait()
            if self.compress == "bf16":
                self.eng.grad[lo:hi].copy_(self._stage[lo:hi])
            if not self._avg and self.world > 1:
                self.eng.grad[lo:hi].div_(self.world)
        self._works.clear()

    def replicas_in_sync(self) -> bool:
        """Cross-rank parameter equality check (the race detector this path gets: every rank must hold bit-identical
        weights after each step)."""
        if self.world == 1:
            return True
        s = self.eng.flat.double().sum().reshape(1)
        a = self.eng.flat.abs().double().sum().reshape(1)
        v = torch.cat([s, a])
        lo, hi = v.clone(), v.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.pg)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.pg)
        return bool(torch.equal(lo, hi))
