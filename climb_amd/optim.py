"""Fused multi-tensor AdamW over the flat parameter buffer (one HIP launch per step).

Semantics are `torch.optim.AdamW` exactly as REF/modeling/vilt.py:205-215 constructs it (decoupled weight decay, bias
correction, per-parameter step counts, parameters whose `.grad` is None are skipped), so it is a drop-in for the
reference's `optimizer.step(); scheduler.step(); optimizer.zero_grad()` sequence (REF train_vqa.py:168-172) and works
with `transformers.get_polynomial_decay_schedule_with_warmup` (a LambdaLR over `param_groups`)."""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _lib


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, host=None):
        if host is None:
            raise ValueError("FusedAdamW needs the engine host of the model whose parameters it updates (model.create_optimizer builds it)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._host = host
        self._eng = None
        self._m = self._v = None
        self._steps = {}            # parameter name -> number of updates so far
        self._seg_group_host = None
        self._seg_start = self._seg_group = None

    def _prepare(self):
        eng = self._host.engine()
        if eng is not self._eng:
            self._eng = eng
            lay = eng.layout
            self._m = torch.zeros(lay.total, dtype=torch.float32, device=eng.device)
            self._v = torch.zeros(lay.total, dtype=torch.float32, device=eng.device)
            segs = lay.segments()
            self._seg_names = [s[0] for s in segs]
            starts = np.array([s[1] for s in segs] + [lay.total], dtype=np.int64)
            self._seg_start = torch.from_numpy(starts).to(eng.device)
            self._seg_start_host = starts
            self._spans, self._nspans, self._nblocks = None, 0, 0
            self._g16_ranges = ()
            self._seg_group = torch.full((len(segs),), -1, dtype=torch.int8, device=eng.device)
            self._seg_group_host = None
            by_id = {id(p): n for n, p in self._host._params.items()}
            self._group_of = {}
            for gi, g in enumerate(self.param_groups):
                for p in g["params"]:
                    n = by_id.get(id(p))
                    if n is None:
                        if not p.requires_grad:         # frozen parameters outside the flat buffer (ViLT-BERT's BERT): torch skips them too
                            continue
                        raise RuntimeError("FusedAdamW: parameter is not part of the bound model (was the model re-created?)")
                    self._group_of[n] = gi
        return eng

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        eng = self._prepare()
        if self._host.ddp is not None:      # backstop: paths that never ran the encoder backward (frozen encoder + autograd heads)
            self._host.ddp.finish()
        params = self._host._params
        combos = {}             # (param group, step count) -> kernel group slot
        seg_group = np.full((len(self._seg_names),), -1, dtype=np.int8)
        for si, n in enumerate(self._seg_names):
            gi = self._group_of.get(n)
            if gi is None:
                continue
            p = params[n]
            if not p.requires_grad or not eng.is_touched(n):      # torch skips parameters whose .grad is None
                continue
            t = self._steps.get(n, 0) + 1
            self._steps[n] = t
            key = (gi, t)
            slot = combos.get(key)
            if slot is None:
                slot = combos[key] = len(combos)
                if slot >= 8:
                    raise RuntimeError("FusedAdamW: more than 8 distinct (group, step) combinations in one step")
            seg_group[si] = slot
        if not combos:
            eng.materialize_dw()
            return loss
        table = np.zeros((len(combos), 8), dtype=np.float32)
        for (gi, t), slot in combos.items():
            g = self.param_groups[gi]
            b1, b2 = g["betas"]
            table[slot] = (g["lr"], g["weight_decay"], b1, b2, g["eps"], 1.0 - b1 ** t, 1.0 - b2 ** t, 0.0)
        fused = set()
        shadow = eng.shadow_ptr()
        # r05: an EWC term parked by the fused training step (engine.park_ewc) is applied inside this step's two passes -- provided they reach EVERY
        # encoder element (the term's value and gradient cover the whole encoder range, REF/cl_algorithms/ewc.py:75-87); else it is written the old way
        fold = eng._ewc_fold
        fold_flat = False
        if fold is not None:
            segs_enc = [si for si, st in enumerate(self._seg_start_host[:-1]) if st < eng.layout.encoder_end]
            if shadow is None or eng._grad_extra or not all(seg_group[si] >= 0 for si in segs_enc):
                eng.apply_parked_ewc()
                fold = None
            else:
                eng._ewc_fold = None
                fold_flat = os.environ.get("CLIMB_AMD_EWC_FOLD", "2") == "2" or getattr(eng, "split", False)          # (the split epilogue has no EWC instantiation)
                if fold_flat:
                    # the term rides in the FLAT pass for every encoder element: the weight gradients are written by the plain launch and the optimizer
                    # is not carried in its epilogue this step (measured, tools/ewc_ab.py: the epilogue is exposed time -- two more operands there cost
                    # more than the separate pass they replace)
                    eng.materialize_dw(keep_parked_ewc=True)
        idx = {n: i for i, n in enumerate(self._seg_names)}          # (also read below when nothing was deferred)
        if eng._dw_deferred:
            # r04: weight-gradient launches held back for this step (the fused training step armed engine.defer_dw): run them with the update in
            # their epilogue for the tensors of the most common (group, step) combination; the flat pass below skips what was updated there
            names = [n for _, plan in eng._dw_deferred for n, w in zip(plan["names"], plan["whole"]) if w]          # (a fused QKV problem is named by its query weight)
            slots = [int(seg_group[idx[n]]) for n in names]
            if (shadow is not None or getattr(eng, "split", False)) and any(sl >= 0 for sl in slots):          # (split mode: the epilogue writes the operand planes itself)
                slot0 = max(set(sl for sl in slots if sl >= 0), key=slots.count)
                row = table[slot0].copy()
                row[7] = 1.0          # gradient scale
                fused = eng.fused_dw_adamw(self._m, self._v, lambda n: int(seg_group[idx[n]]) == slot0, row, ewc=fold)
                for n in fused:
                    seg_group[idx[n]] = -1
            else:
                eng.materialize_dw()
        g16 = eng._g16                 # data parallel: averaged gradients still in the reducer's 16-bit payload buffer (parallel.finish(defer_uncast=True))
        g16_ranges = tuple(g16["ranges"]) if g16 else ()
        if self._seg_group_host is None or not np.array_equal(seg_group, self._seg_group_host) or g16_ranges != self._g16_ranges:
            if self._seg_group_host is None or not np.array_equal(seg_group, self._seg_group_host):
                self._seg_group.copy_(torch.from_numpy(seg_group))
                self._seg_group_host = seg_group
            self._g16_ranges = g16_ranges
            # maximal runs of tensors the flat pass updates, as { first element, elements, first 1024-element block, gradient source } (csrc/optim.hip::
            # adamw_spans_kernel); a run is cut where it crosses into / out of a range whose gradient lives in the 16-bit payload buffer (source 1)
            starts = self._seg_start_host
            runs = []
            for si in np.flatnonzero(seg_group >= 0):
                a, b = int(starts[si]), int(starts[si + 1])
                if runs and runs[-1][1] == a:
                    runs[-1][1] = b
                else:
                    runs.append([a, b])
            spans, nb = [], 0
            for a, b in runs:
                cuts = sorted({a, b} | {x for lo, hi in g16_ranges for x in (lo, hi) if a < x < b})
                for c0, c1 in zip(cuts[:-1], cuts[1:]):
                    src = int(any(lo <= c0 and c1 <= hi for lo, hi in g16_ranges))
                    spans.append([c0, c1 - c0, 0, src])
            for sp in spans:
                sp[2] = nb
                nb += (sp[1] + 1023) // 1024
            self._spans = torch.tensor(spans, dtype=torch.int64, device=eng.device).reshape(-1) if spans else None
            self._nspans, self._nblocks = len(spans), nb
        # The pass clears the gradients it consumes when that leaves the WHOLE buffer zero: every range the backward wrote is updated here (the
        # matrices updated in the weight-gradient epilogue were never written), nothing else is parked in the buffer (EWC term, accumulated sums)
        # and no reducer or loss scale is in play.  The optimizer.zero_grad() that follows (REF/.../train_vqa.py:170) then has nothing to fill.
        clean = bool(fused) and not eng._grad_extra and self._host.ddp is None and eng.loss_scale == 1.0 and not eng._dw_deferred
        if fold is not None and fold_flat and self._host.ddp is None and eng.loss_scale == 1.0:
            clean = True          # (the plain launch's sums are this step's gradients: the flat pass consumes -- and clears -- every range, checked below)
        if clean:
            consumed = seg_group >= 0
            consumed[[idx[n] for n in fused]] = True
            starts = self._seg_start_host
            for lo, hi in eng.touched:
                i0, i1 = int(np.searchsorted(starts, lo, "right")) - 1, int(np.searchsorted(starts, hi, "left"))
                if not consumed[i0:i1].all():
                    clean = False
                    break
        if self._nspans:
            args = (eng.flat, eng.grad, self._m, self._v, shadow, self._spans, self._nspans, self._nblocks, self._seg_start,
                    self._seg_group, len(self._seg_names), table.ctypes.data, len(combos), 1.0, 1 if clean else 0,
                    g16["stage"] if g16 else None, g16["scale"] if g16 else 1.0)
            if fold is not None:
                _lib.call("climb_adamw_spans_ewc", *args, fold["star"], fold["fisher"], eng.layout.encoder_end, fold["lam"], fold["loss"],
                          torch.cuda.current_stream().cuda_stream)
            else:
                _lib.call("climb_adamw_spans", *args, torch.cuda.current_stream().cuda_stream)
        eng._g16 = None                # consumed
        eng._grad_clean = clean
        # (r06) which tensors changed at all: the transposed shadows of everything else stay as they are (a frozen base under adapters)
        updated = set(fused) | {n for si, n in enumerate(self._seg_names) if seg_group[si] >= 0}
        # (split mode: the flat pass writes no planes -- the shadows are "fresh" exactly where the epilogue wrote them; refresh_shadow() re-splits the rest)
        eng.params_updated(shadow_fresh=(shadow is not None) or (getattr(eng, "split", False) and bool(fused)), t_fresh=fused, updated=updated)
        return loss

    # ---- checkpointing: the moments and per-parameter step counts live in flat buffers outside `self.state`, so the inherited
    # state_dict() would serialise nothing and a resume would silently restart the moments and the bias correction.
    def state_dict(self):
        sd = super().state_dict()
        if self._m is not None and self._steps:
            sd["climb_amd_flat"] = dict(m=self._m.detach().clone(), v=self._v.detach().clone(), steps=dict(self._steps),
                                        total=int(self._eng.layout.total))
        return sd

    def load_state_dict(self, state_dict):
        flat = state_dict.get("climb_amd_flat")
        super().load_state_dict({k: v for k, v in state_dict.items() if k != "climb_amd_flat"})
        if flat is None:
            self._steps = {}
            if self._m is not None:
                self._m.zero_()
                self._v.zero_()
            return
        eng = self._prepare()
        if int(flat["total"]) != int(eng.layout.total):
            raise ValueError(f"FusedAdamW.load_state_dict: optimizer state was saved for a parameter layout of {flat['total']} elements, "
                             f"this model has {eng.layout.total} (different task heads / adapters?)")
        self._m.copy_(flat["m"])
        self._v.copy_(flat["v"])
        self._steps = dict(flat["steps"])

    def zero_grad(self, set_to_none: bool = True):
        """One memset of the flat gradient buffer; every `.grad` becomes None (torch's set_to_none=True behaviour)."""
        self._host.engine()
        self._host.drop_grads()
