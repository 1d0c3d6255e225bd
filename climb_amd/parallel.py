"""Data parallelism for the ViLT step: one process per GPU, full replica, gradient all-reduce over RCCL/xGMI overlapped
with the backward.  The reference has no distributed code at all; this is new design, not a counterpart.

Why it is shaped like this on MI355X
  * xGMI is point-to-point (7 links x ~153 GB/s per GPU), so collectives are per-link bound: few LARGE messages beat
    many small ones.  The flat gradient buffer is laid out in forward order, so every encoder layer is ONE contiguous
    28 MB fp32 range: the engine reports `(lo, hi)` the moment a layer's weight gradients are final and that range is
    all-reduced in place -- no bucket copy-in/copy-out, no per-tensor calls.
  * Only TRAINABLE ranges are reported (frozen layers, a frozen base under adapters: nothing on the wire).  Fragments smaller
    than a bucket that are not contiguous (the 24 adapter blocks of a step: 7 MB in all) are PACKED into one staging buffer and
    travel as one collective instead of 24 latency-bound ones.
  * The collective is enqueued with `async_op=True`: torch's RCCL process group runs it on its own HIP stream after an
    event on the compute stream, so layer i's all-reduce rides under layer i-1's backward GEMMs.  `CLIMB_AMD_DP_OVERLAP=0`
    defers every collective to `finish()` (after the backward): the measurement knob for the question DESIGN.md §8 leaves open --
    a concurrent HBM stream slows the latency-bound GEMM k-loops, so overlap is not automatically a win on this chip.
  * `finish()` makes the compute stream wait for every outstanding collective; only then are the EWC penalty gradient
    (identical on every rank, so it must NOT be summed) and the fused AdamW enqueued.  The fused optimizer calls `finish()` itself
    as a backstop, so paths that never reach the encoder backward (frozen encoder on the reference-style autograd path) are reduced too.
  * ReduceOp.AVG on RCCL; SUM followed by a scale on backends without AVG (gloo, used by the CPU tests).
  * Payload: fp32 (bit-faithful averaging, the parity default and -- r05 -- the default of the IEEE-half build, whose half payload can overflow:
    see attach()) or 16 bit (`compress="bf16"`, the default of the bf16 throughput mode; SURVEY.md §8(e) "prefer bf16 payload"): the range is cast into a persistent bf16 staging buffer laid out like the
    gradient buffer (`climb_cast_bf16`), reduced there, and cast back with the averaging scale folded in
    (`climb_uncast_bf16_scale`).  Halves the bytes every xGMI link carries (480 -> 240 MB per step) for two extra streaming
    passes over the gradients; the rounding (2^-9 relative per element, once) is below the bf16 GEMM noise already in those
    gradients.  `CLIMB_AMD_DP_COMPRESS=none|bf16` overrides.
  * Learning-rate / schedule rule under N ranks: the reference's hyper-parameters are tuned for ITS batch size, so a
    data-parallel run keeps the GLOBAL batch equal to the reference's (`--batch_size` is the global batch, each rank loads
    batch_size / N), leaves lr, warm-up ratio and steps per epoch unchanged, and averages (not sums) gradients.  `bench.py` instead
    holds the per-GPU batch fixed (weak scaling, as the driver's contract asks) and says so in its JSON line.
"""
from __future__ import annotations

import contextlib
import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from . import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


_DEFER_UNCAST = os.environ.get("CLIMB_AMD_DP_DEFER_UNCAST", "1") != "0"      # A/B knob: 0 = always cast the averaged payload back into the gradient buffer


def _cast_to_bf16(dst: torch.Tensor, src: torch.Tensor):
    if src.is_cuda:
        _lib.call("climb_cast_bf16", src, dst, src.numel(), _stream())
    else:                      # CPU tensors exist only in the gloo tests of this module
        dst.copy_(src)


def _uncast_scaled(dst: torch.Tensor, src: torch.Tensor, scale: float):
    if src.is_cuda:
        _lib.call("climb_uncast_bf16_scale", src, dst, src.numel(), scale, _stream())
    else:
        dst.copy_(src)
        if scale != 1.0:
            dst.mul_(scale)


def _scale(x: torch.Tensor, scale: float):
    if scale == 1.0:
        return
    if x.is_cuda:
        _lib.call("climb_scale_f32", x, x.numel(), scale, _stream())
    else:
        x.mul_(scale)


class GradientAllReducer:
    def __init__(self, model=None, process_group=None, min_bucket_elems: int = 4 * 1024 * 1024, broadcast: bool = True,
                 compress: Optional[str] = None, overlap: Optional[bool] = None):
        self.pg = process_group
        self.compress = compress            # None = decide at attach() from the engine's precision
        self.h16_payload, self.payload_dtype = False, torch.float32
        self._stage = None                  # bf16 payload staging, laid out like the gradient buffer
        self._packs: List[torch.Tensor] = []   # packed payloads of non-contiguous small fragments (payload dtype), recycled every step
        self._packs_used = 0
        self.enabled = True                 # False inside suspended(): ranges are not reduced (the replicated Fisher pass)
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.min_bucket = min_bucket_elems
        self.eng = None
        self.overlap = (os.environ.get("CLIMB_AMD_DP_OVERLAP", "1") != "0") if overlap is None else bool(overlap)
        # CUs left to RCCL while collectives run UNDER the backward (overlap mode only).  The persistent GEMMs launch one 8-wave / 256-VGPR workgroup
        # per CU, which cannot share a CU with an RCCL workgroup: with k CUs held by a collective, k of the 256 workgroups of a multi-round launch
        # only start when the others have finished ALL their tiles (the launch takes twice as long).  With a reserve the engine sizes those grids
        # to the CUs that are free (tiles spread evenly) from the first collective of a step until finish().  0 = off; bench.py --gpus N tries it
        # in its warm-up next to plain overlap and deferred collectives and keeps the fastest.
        self.reserve_cus = int(os.environ.get("CLIMB_AMD_DP_RESERVE_CUS", "0"))
        self._reserved = False
        self._in_finish = False
        self._works: List[Tuple[object, List[Tuple[int, int]], Optional[torch.Tensor]]] = []
        self._small: List[Tuple[int, int]] = []
        self._deferred: List[Tuple[int, int]] = []
        self._avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        self.bytes_reduced = 0
        self.collectives = 0
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
        fp16_lib = getattr(eng, "h16", None) == "fp16"
        # 16-bit payload: "bf16" on the bf16 build; on the IEEE-half build the cast kernels produce half, whose range only holds the gradients
        # while they still carry the engine's loss scale -- "fp16": the engine hands ranges over BEFORE unscaling them (takes_scaled) and
        # finish() divides the scale out together with the averaging factor
        # r05: the half payload is OPT-IN (CLIMB_AMD_DP_COMPRESS=fp16 / compress="fp16"), the half build defaults to the fp32 payload.  The engine's loss
        # scale puts |d(logits)| near 2^7, but a weight gradient sums over a batch's tokens: bench.py's two-rank run on this build (8 sequences per rank,
        # scale 2^10) carried 3.5e4 in the position-embedding gradient of ONE rank -- their sum left half's range (6.9e4 > 65504), the average came back
        # inf and the replicas NaN (found by tests/test_gpu_rccl.py::test_bench_two_ranks_on_one_gpu_through_gloo on the half build).  bf16 has fp32's
        # range: the bf16 build keeps its 16-bit payload.
        if mode is None:
            mode = ("none" if fp16_lib else "bf16") if getattr(eng, "precision", "fp32") == "bf16" else "none"
        if mode not in ("none", "bf16", "fp16"):
            raise ValueError(f"unknown gradient compression {mode!r}")
        if (mode == "bf16" and fp16_lib) or (mode == "fp16" and not fp16_lib):
            raise ValueError(f"gradient compression {mode!r} needs the {mode} build of the library (this one casts to "
                             f"{'IEEE half' if fp16_lib else 'bf16'})")
        self.h16_payload = mode in ("bf16", "fp16")
        self.payload_dtype = (torch.float16 if mode == "fp16" else torch.bfloat16) if self.h16_payload else torch.float32
        self.compress = mode
        self._stage = None
        self._packs, self._packs_used = [], 0

    @property
    def takes_scaled(self) -> bool:
        """True when ranges must reach on_ready() still multiplied by the engine's loss scale (half payload, more than one rank: with one
        rank nothing is cast and the engine unscales as usual)."""
        # (not while suspended(): the replicated Fisher pass reduces nothing, so nobody would divide the scale out again -- the engine must unscale
        # those gradients itself before fisher_accumulate() squares them)
        return self.enabled and self.compress == "fp16" and self.world > 1

    def begin(self):
        self._works.clear()
        self._small.clear()
        self._deferred.clear()
        self._packs_used = 0            # finish() of the previous step waited for every collective: its pack buffers are free again

    @contextlib.contextmanager
    def suspended(self):
        """Inside: backward passes reduce nothing.  For passes every rank runs on IDENTICAL data (EWC's Fisher estimate: REF/cl_algorithms/
        ewc.py:56-68 accumulates gradients across batches in order and cannot be sharded without changing the result)."""
        prev, self.enabled = self.enabled, False
        try:
            yield self
        finally:
            self.enabled = prev

    # ------------------------------------------------------------------ collectives
    def _reserve(self, on: bool):
        if on == self._reserved or self.eng is None or not hasattr(self.eng, "set_cu_reserve"):
            return
        self.eng.set_cu_reserve(self.reserve_cus if on else 0)
        self._reserved = on

    def _reduce(self, payload: torch.Tensor, ranges, packed):
        if self.reserve_cus > 0 and not self._in_finish:      # a collective is about to run next to the rest of the backward
            self._reserve(True)
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        work = dist.all_reduce(payload, op=op, group=self.pg, async_op=True)
        self._works.append((work, ranges, packed))
        self.bytes_reduced += payload.numel() * payload.element_size()
        self.collectives += 1

    def _launch(self, lo: int, hi: int):
        """one contiguous range, reduced in place (fp32) or through the like-shaped bf16 staging buffer"""
        if self.world == 1:
            return
        chunk = self.eng.grad[lo:hi]
        if self.h16_payload:
            if self._stage is None:
                self._stage = torch.empty(self.eng.grad.numel(), dtype=self.payload_dtype, device=self.eng.grad.device)
            _cast_to_bf16(self._stage[lo:hi], chunk)
            chunk = self._stage[lo:hi]
        self._reduce(chunk, [(lo, hi)], None)

    def _launch_packed(self, ranges: List[Tuple[int, int]]):
        """several non-contiguous small ranges as ONE collective over a packed copy"""
        if self.world == 1:
            return
        n = sum(hi - lo for lo, hi in ranges)
        dt = self.payload_dtype
        k = self._packs_used            # the k-th packed collective of a step reuses the k-th buffer of the pool (no allocation in the step)
        if k == len(self._packs):
            self._packs.append(None)
        pk = self._packs[k]
        if pk is None or pk.numel() < n or pk.dtype != dt:
            pk = self._packs[k] = torch.empty(max(n, self.min_bucket), dtype=dt, device=self.eng.grad.device)
        self._packs_used += 1
        buf = pk[:n]
        at = 0
        for lo, hi in ranges:
            if self.h16_payload:
                _cast_to_bf16(buf[at:at + hi - lo], self.eng.grad[lo:hi])
            else:
                buf[at:at + hi - lo].copy_(self.eng.grad[lo:hi])
            at += hi - lo
        self._reduce(buf, list(ranges), buf)          # in flight until finish(); the next packed bucket of this step takes the next buffer

    def _flush_small(self):
        if not self._small:
            return
        merged: List[List[int]] = []
        for lo, hi in sorted(self._small):
            if merged and merged[-1][1] == lo:
                merged[-1][1] = hi
            else:
                merged.append([lo, hi])
        self._small.clear()
        if len(merged) == 1:
            self._launch(merged[0][0], merged[0][1])
        else:
            self._launch_packed([(a, b) for a, b in merged])

    def on_ready(self, lo: int, hi: int):
        """Ranges arrive in backward order (head, final norm + pooler, layer 11 ... layer 0, embeddings; only trainable sub-ranges).
        A range worth a collective goes at once, in place; smaller ones wait for neighbours and go together."""
        if hi <= lo or not self.enabled:
            return
        if not self.overlap:
            self._deferred.append((lo, hi))
            return
        self._dispatch(lo, hi)

    def _dispatch(self, lo: int, hi: int):
        if hi - lo >= self.min_bucket:
            # a pending small neighbour that touches this range rides with it
            for k, (slo, shi) in enumerate(self._small):
                if shi == lo or slo == hi:
                    lo, hi = min(lo, slo), max(hi, shi)
                    self._small.pop(k)
                    break
            self._launch(lo, hi)
        else:
            self._small.append((lo, hi))
            if sum(b - a for a, b in self._small) >= self.min_bucket:
                self._flush_small()

    def finish(self, defer_uncast: bool = False):
        """Block the compute stream on every outstanding collective (no host sync on RCCL) and put the averaged gradients back.
        `defer_uncast` (r04; the fused step passes it when the caller named its FusedAdamW as the next reader of the gradients and no EWC term has to
        be added to them): contiguous ranges that went through the 16-bit staging buffer are NOT cast back -- the engine is told where the averaged
        payload lives (`eng._g16`) and the optimizer's flat pass reads it from there with the scale folded in (`climb_adamw_spans`), so the un-cast
        pass (6 B per parameter) and the fp32 re-read disappear.  Anything else that wants those gradients calls `eng.materialize_g16()` first."""
        if not self.enabled:
            return
        pending16 = []
        self._in_finish = True
        try:
            for lo, hi in self._deferred:
                self._dispatch(lo, hi)
            self._deferred.clear()
            self._flush_small()
        finally:
            self._in_finish = False
        scale = 1.0 if (self._avg or self.world == 1) else 1.0 / self.world
        if self.takes_scaled:            # the half payload was cast from gradients that still carried the loss scale
            scale /= float(getattr(self.eng, "loss_scale", 1.0))
        for work, ranges, packed in self._works:
            work.wait()
            if packed is not None:
                at = 0
                for lo, hi in ranges:
                    src = packed[at:at + hi - lo]
                    if self.h16_payload:
                        _uncast_scaled(self.eng.grad[lo:hi], src, scale)
                    else:
                        self.eng.grad[lo:hi].copy_(src)
                        _scale(self.eng.grad[lo:hi], scale)
                    at += hi - lo
            else:
                lo, hi = ranges[0]
                if self.h16_payload:
                    if defer_uncast and self._stage.is_cuda and _DEFER_UNCAST:
                        pending16.append((lo, hi))
                    else:
                        _uncast_scaled(self.eng.grad[lo:hi], self._stage[lo:hi], scale)
                else:
                    _scale(self.eng.grad[lo:hi], scale)
        if pending16:
            self.eng._g16 = dict(stage=self._stage, scale=float(scale), ranges=sorted(pending16))
        self._works.clear()
        self._reserve(False)            # nothing of RCCL's is on the chip any more: the persistent grids take every CU again
        self._packs_used = 0            # every collective was waited for: the pack buffers are free (also on paths that never call begin())

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


# ---------------------------------------------------------------------------------------------------------------- job-level helpers
def rank_world(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def barrier(group=None):
    """No-op outside a data-parallel job."""
    if rank_world(group)[1] > 1:
        dist.barrier(group=group)


def ensure_reducer(model) -> Optional[GradientAllReducer]:
    """What a trainer's train() does first under N > 1 ranks: attach the gradient all-reducer to the model's engine (once; later tasks and
    re-bound engines keep it) after broadcasting rank 0's parameters.  Returns None in a single-process run."""
    if rank_world()[1] == 1:
        return None
    host = model._host
    if host.ddp is None:
        GradientAllReducer(model)
    return host.ddp


def init_data_parallel(backend: Optional[str] = None):
    """One process per GPU under torchrun (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment): bind this process to its GPU
    and join the job.  Returns (rank, world, device).  RCCL ("nccl") on GPUs; gloo is for the CPU tests."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local % torch.cuda.device_count())
        device = torch.device("cuda", local % torch.cuda.device_count())
    else:
        device = torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": device} if (device.type == "cuda" and (backend or "nccl") == "nccl") else {}
        dist.init_process_group(backend or ("nccl" if device.type == "cuda" else "gloo"), rank=rank, world_size=world, **kw)
    return rank, world, device


_real_torch_save = None
_real_makedirs = None
_real_json_dump = None
COLLECTIVE_JSON = {"results.json", "eval_results.json", "lowshot_results.json"}      # a launcher may register more names
IO_LOG_MAX = 4096    # entries kept (long runs: the log is for tests and post-mortems, not an unbounded record)
io_log = []          # (kind, path, wrote) per collective file operation of this process: what tests assert "one writer per file" on


def rank0_only_io():
    """Makes the reference driver's file IO safe for N processes running the same code (prefer the context manager `single_writer_io()`, which
    also takes the patches away again).  The driver
      * saves checkpoints with `torch.save` from its only process (REF/train/train_upstream_continual_learning.py:261-267): under N ranks the
        replicas are bit-identical, so one copy is enough -- on ranks > 0 `torch.save` of a path is a no-op, and every rank leaves the call through a
        barrier, so the file exists for all ranks once any of them has moved on.  (Consequence: while this is installed a path-based torch.save is a
        COLLECTIVE -- a save only some ranks make would deadlock.)
      * creates directories with `if not isdir: makedirs` (:118-119, :261-262), which two processes race through: `os.makedirs` tolerates an
        existing directory;
      * writes results.json / eval_results.json with `json.dump(obj, open(path, "w"))` from every rank (:269-277, :327) -- identical content, the same
        path, concurrent writers: a torn file is possible.  `json.dump` into a file opened for writing becomes: barrier (every rank has opened -- and
        truncated -- the path by now), rank 0 alone writes a temporary file next to it and `os.replace`s it over the path, barrier.  The handles the
        other ranks hold point at the old inode; nothing they do can touch the new file.
    Idempotent; `restore_io()` undoes it."""
    global _real_torch_save, _real_makedirs, _real_json_dump
    rank, world = rank_world()
    if world == 1 or _real_torch_save is not None:
        return
    import json
    _real_torch_save, _real_makedirs, _real_json_dump = torch.save, os.makedirs, json.dump

    def makedirs(name, mode=0o777, exist_ok=False):
        return _real_makedirs(name, mode, exist_ok=True)
    os.makedirs = makedirs

    def save(obj, f, *a, **kw):
        is_path = isinstance(f, (str, os.PathLike))
        if rank == 0 or not is_path:
            _real_torch_save(obj, f, *a, **kw)
        if is_path:
            if len(io_log) < IO_LOG_MAX:
                io_log.append(("torch.save", os.fspath(f), rank == 0))
            dist.barrier()
    torch.save = save

    def dump(obj, fp, *a, **kw):
        path = getattr(fp, "name", None)
        # only the drivers' own results files are written collectively (REF/train/train_upstream_continual_learning.py:277,327,
        # train_lowshot_multimodal.py:183,233): any other json.dump -- run metadata, a library's or one rank's own log file -- need not be called by
        # every rank the same number of times, and a barrier inside it would dead-lock or mis-pair
        if not (isinstance(path, str) and "w" in getattr(fp, "mode", "") and os.path.isfile(path) and os.path.basename(path) in COLLECTIVE_JSON):
            return _real_json_dump(obj, fp, *a, **kw)
        dist.barrier()                                          # every rank is past its open(path, "w")
        if rank == 0:
            tmp = f"{path}.tmp.{os.getpid()}"
            with open(tmp, "w") as t:
                _real_json_dump(obj, t, *a, **kw)
            os.replace(tmp, path)
        if len(io_log) < IO_LOG_MAX:
            io_log.append(("json.dump", path, rank == 0))
        dist.barrier()
    json.dump = dump


def restore_io():
    global _real_torch_save, _real_makedirs, _real_json_dump
    if _real_torch_save is not None:
        import json
        torch.save, os.makedirs, json.dump = _real_torch_save, _real_makedirs, _real_json_dump
        _real_torch_save = _real_makedirs = _real_json_dump = None


@contextlib.contextmanager
def single_writer_io():
    """`with parallel.single_writer_io(): run the driver` -- the patches of rank0_only_io() exist for the duration of the driver call only."""
    rank0_only_io()
    try:
        yield
    finally:
        restore_io()
