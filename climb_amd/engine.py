"""Host-side step engine: owns the flat parameter / gradient buffers and the activation workspace in HBM and
sequences the C-ABI HIP launchers for the ViLT forward, backward and update.  PyTorch is used for device memory,
streams and (elsewhere) torch.distributed only; every FLOP and every byte of the step goes through libclimb_hip.so.

Three arithmetic modes share all of this code:
  * "fp32": exact-fp32 matrix-core GEMMs (v_mfma_f32_32x32x2_f32) -- the parity mode (<= 1e-3 rel vs the CPU reference,
            argmax exact) of BASELINE.json's north_star
  * "bf16": bf16 MFMA operands, fp32 accumulation / statistics / residual stream / master weights -- the throughput mode
            (BASELINE.json configs[1])
  * "bf16x3" (r06): the fp32 mode's data flow (fp32 activations, statistics, attention, exact GELU) with every encoder GEMM on SPLIT operands --
            (hi, lo) bf16 plane pairs, three MFMA products per k-step (csrc/split.hip): inside the 1e-3 / argmax bar at a third of the 16-bit
            MFMA rate instead of a sixteenth
"""
from __future__ import annotations

import math
import os
from typing import Callable, Dict, List, Optional

import torch

from . import _lib
from .layout import ENC, FlatLayout, VILT_CFG, TASK_ARITH

F32, BF16, SPLIT = 0, 1, 2
_FUSED_ADAPTER = os.environ.get("CLIMB_AMD_FUSED_ADAPTER", "1") != "0"       # measurement knob: 0 = the two skinny GEMMs
# weight gradients of the encoder layers as grouped launches (csrc/gemm_bf16_tnp.hip: gemm_bf16_tn_grouped_kernel): layers per launch.
# "0" = off (one split GEMM + reduce per weight, the r02 path); default: all layers in one launch, 4 per launch under a data-parallel hook
# (ranges must become ready in a few chunks for the all-reduce to overlap the rest of the backward)
_DW_GROUP = os.environ.get("CLIMB_AMD_DW_GROUP")
_NT_GRID_BEFORE_RESERVE = 0        # library-wide persistent NT grid (option 9) in force before the first CU reserve
import weakref
_NT_RESERVING = weakref.WeakSet()  # engines that currently hold a reserve (weak: an engine that dies holding one -- a model re-created between tasks -- drops out)


def _restore_nt_grid_if_unreserved(cell):
    """finalizer of an engine that DIED holding a CU reserve (cell[0] > 0): once nobody reserves any more, the library-wide persistent NT grid goes back to
    what was in force before the first reserve"""
    try:
        if cell[0] > 0 and not len(_NT_RESERVING):
            _lib.call("climb_set_option", 9, _NT_GRID_BEFORE_RESERVE)
    except Exception:
        pass

_RED_BATCH = os.environ.get("CLIMB_AMD_RED_BATCH", "1") != "0"          # measurement knob: 0 = one reduce launch per LayerNorm backward
_UNSCALE_MODE = os.environ.get("CLIMB_AMD_FP16_UNSCALE", "end")      # measurement knob: "range" (per finished range), "end" (one pass), "none" (timing only)
EPI_NONE, EPI_GELU, EPI_RESID, EPI_DGELU, EPI_TANH, EPI_SILU, EPI_DSILU, EPI_RESID2, EPI_GELUD, EPI_MUL = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9
EPI_GELU_SP, EPI_DGELU_SP = 10, 11
# r06, split mode: GELU / x GELU' as epilogues of the split NT launch (csrc/gemm_bf16_nt4.hip, erf form to 1.5e-7) instead of a pass of their own over the fp32
# result.  Measured (profiles/r06_split_act_ab.txt): the two launches 148 -> 191 / 180 us (each unit's ~90 VALU operations sit in ONE MFMA slot of the window that
# drains its block: exposed), the 50 + 79 us passes gone: step 23.38 -> 22.77 ms.  On by default; "0" = the separate passes with the exact erff forms.
SPLIT_FUSED_ACT = os.environ.get("CLIMB_AMD_SPLIT_FUSED_ACT", "1") != "0"
# r05: what the MLP's up-projection saves for the backward in the 16-bit modes.  "deriv" (default): gelu'(pre-activation), computed in the forward
# epilogue from the sigmoid the activation needs anyway (5 more operations per element), so that the backward's epilogue is ONE multiply per element
# (EPI_GELUD / EPI_MUL); "pre": the pre-activation itself, the derivative evaluated in the backward (EPI_GELU / EPI_DGELU; the r01 - r04 scheme and what
# the fp32 parity mode does).  Same arithmetic up to the 16-bit rounding of the saved tensor.
GELU_SAVE = os.environ.get("CLIMB_AMD_GELU_SAVE", "deriv")


# r04: the pooler / task-head products on csrc/heads.hip (0 = the r01-r03 launches: split-K GEMMs with atomics + separate activations)
SKINNY = os.environ.get("CLIMB_AMD_SKINNY_HEADS", "1") != "0"


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _round_up(x, m):
    return (x + m - 1) // m * m


_TN_WS = {}


def tn_workspace(device):
    """Scratch for the weight-gradient kernel's split partial sums (plain stores + one reduce launch instead of fp32 atomics).  The
    library keeps the registered pointer for the life of the process, so the buffer is a per-device singleton that is never freed
    (64 MB: covers every ViLT-B weight shape at 12288 tokens); one process drives one GPU."""
    key = str(torch.device(device))
    if key not in _TN_WS:
        _TN_WS[key] = torch.empty(64 * 1024 * 1024 // 4, dtype=torch.float32, device=device)
    _lib.call("climb_set_tn_workspace", _TN_WS[key], _TN_WS[key].numel() * 4)
    return _TN_WS[key]


_NT_WS = {}


def nt_workspace(device):
    """Scratch of the split-along-K NT GEMMs (csrc/gemm_bf16_ntp.hip: partial accumulator tiles handed between workgroups, 38 MB).  Like
    tn_workspace: a per-device singleton the library keeps a pointer to for the life of the process."""
    key = str(torch.device(device))
    if key not in _NT_WS:
        _NT_WS[key] = torch.empty(_lib.query("climb_nt_workspace_bytes") // 4, dtype=torch.float32, device=device)
    _lib.call("climb_set_nt_workspace", _NT_WS[key], _NT_WS[key].numel() * 4)
    return _NT_WS[key]


class Workspace:
    """Activation + scratch buffers for one (B, T) shape; allocated once, reused every step."""

    def __init__(self, eng: "ViltEngine", B: int, T: int, gh: int = None, gw: int = None, nseq: int = None):
        cfg = eng.cfg
        dev = eng.device
        H, Fd, L, nh = cfg["hidden"], cfg["ffn"], cfg["layers"], cfg["heads"]
        self.B, self.T = B, T
        g0 = cfg["image"] // cfg["patch"]
        self.gh, self.gw = gh or g0, gw or g0         # patch canvas of the (padded) batch
        self.NP = self.gh * self.gw                   # canvas patches (rows of the patch projection)
        self.NS = nseq or self.NP                     # patch rows of a SEQUENCE: the canvas, or (compact) the largest valid count of a sample
        self.compact = self.NS != self.NP
        self.S = T + 1 + self.NS
        self.S_pad = _round_up(self.S, 32)
        self.M = B * self.S_pad
        M = self.M
        adt = torch.float32 if eng.precision == "fp32" else eng.t16
        f32 = torch.float32
        sp = eng.split

        def buf(shape, dt=f32):
            return torch.empty(shape, dtype=dt, device=dev)

        def opbuf(rows, cols):
            """a tensor that only GEMMs read: the operand dtype, or (split mode) a (hi, lo) pair of 16-bit planes [2][rows][cols]"""
            return buf((2, rows, cols), eng.t16) if sp else buf((rows, cols), adt)
        self.key_bias = buf((B, self.S_pad))
        self.img_type = torch.empty((B,), dtype=torch.int32, device=dev)
        self.dims = torch.empty((B, 2), dtype=torch.int32, device=dev)     # valid patch extent per sample (variable resolution)
        self.tmean, self.trstd = buf((B * T,)), buf((B * T,))
        self.a_patch = buf((B * self.NP, cfg["channels"] * cfg["patch"] ** 2), adt)
        self.proj = buf((B * self.NP, H))
        self.x = [buf((M, H)) for _ in range(L + 1)]          # residual stream at every layer boundary (fp32)
        self.h1 = [buf((M, H)) for _ in range(L)]
        self.xn = [opbuf(M, H) for _ in range(L)]
        self.hn = [opbuf(M, H) for _ in range(L)]
        self.qkv = [buf((M, 3 * H), adt) for _ in range(L)]
        self.ctx = [buf((M, H), adt) for _ in range(L)]
        self.u = [buf((M, Fd), adt) for _ in range(L)]
        self.a = [opbuf(M, Fd) for _ in range(L)]
        if sp:      # split twins of the fp32 tensors that something besides a GEMM reads too (attention output / gradient, im2col, d(projection))
            self.a_patch_s = opbuf(B * self.NP, cfg["channels"] * cfg["patch"] ** 2)
            self.ctx_s = [opbuf(M, H) for _ in range(L)]
            self.dqkv_s = opbuf(M, 3 * H)
            self.dproj_s = opbuf(B * self.NP, H)
            self.du_f = buf((M, Fd))          # d(gelu output) before x gelu'(u) (the input-gradient GEMM's fp32 result)
        self.mean1 = [buf((M,)) for _ in range(L)]
        self.rstd1 = [buf((M,)) for _ in range(L)]
        self.mean2 = [buf((M,)) for _ in range(L)]
        self.rstd2 = [buf((M,)) for _ in range(L)]
        self.lse = [buf((B, nh, self.S_pad)) for _ in range(L)]
        self.clsn, self.fmean, self.frstd = buf((B, H)), buf((B,)), buf((B,))
        self.pooled = buf((B, H))
        # the last layer on its [CLS] rows only (ViltEngine.cls_only_last): compact [B, .] twins of h1 / hn / u / a / x_L and of the backward's
        # d(x_L) / d(u) / d(hn) / d(h1) operands
        self.h1c, self.xLc, self.mean2c, self.rstd2c = buf((B, H)), buf((B, H)), buf((B,)), buf((B,))
        self.hnc, self.uc, self.ac = buf((B, H), adt), buf((B, Fd), adt), buf((B, Fd), adt)
        self.dyc, self.duc, self.dhnc, self.dhcc = buf((B, H), adt), buf((B, Fd), adt), buf((B, H), adt), buf((B, H), adt)
        # backward scratch (shared by all layers)
        self.dres = buf((M, H))
        self.dres_c = opbuf(M, H) if sp else (self.dres if eng.precision == "fp32" else buf((M, H), adt))
        self.du = opbuf(M, Fd)
        self.dhn = buf((M, H), adt)
        self.dctx = buf((M, H), adt)
        self.dqkv = buf((M, 3 * H), adt)
        self.dxn = buf((M, H), adt)
        self.delta = buf((B, nh, self.S_pad))
        self.dproj = buf((B * self.NP, H), adt)
        self.dclsn, self.dpre = buf((B, H)), buf((B, H))
        lnb = _lib.query("climb_layernorm_bwd_rows_per_block")
        csr = _lib.query("climb_colsum_rows_per_block")
        etb = _lib.query("climb_embed_text_bwd_rows_per_block")
        npart = max(((M + lnb - 1) // lnb) * 3 * H, ((M + csr - 1) // csr) * max(Fd, 3 * H),
                    (self.NP + 1) * 3 * H, ((B * T + etb - 1) // etb) * 3 * H)
        self.part = buf((npart,))
        self.dpre = buf((B * T, H))
        self.part2 = buf((T * 2 * H,))
        # Houlsby adapters (two per layer): sub-layer output y, bottleneck pre-activation z and s = silu(z)
        self.has_adapters = bool(eng.layout.adapters)
        if self.has_adapters:
            r = max(eng.layout.adapters.values())
            self.r = r
            self.ya = [buf((M, H), adt) for _ in range(L)]
            self.yo = [buf((M, H), adt) for _ in range(L)]
            self.za = [buf((M, r), adt) for _ in range(L)]
            self.sa = [buf((M, r), adt) for _ in range(L)]
            self.zo = [buf((M, r), adt) for _ in range(L)]
            self.so = [buf((M, r), adt) for _ in range(L)]
            self.dz = buf((M, r), adt)
            self.dy = buf((M, H), adt)
            if sp:
                self.dy_s = opbuf(M, H)          # d(sub-layer output) as GEMM operand planes
        self.ones = torch.ones((max(B, 8),), dtype=f32, device=dev)
        self.dw_plans = {}           # grouped weight-gradient launches: problem / item tables in HBM, built once per set of GEMMs
        self.red_plans = {}          # batched column reductions: segment tables in HBM
        self.part_l = None
        self.dx_c = self.dh_c = self.du_l = self.dqkv_l = None

    def ensure_deferred(self, eng: "ViltEngine"):
        """Per-LAYER copies of the four weight-gradient operands the backward produces (d(x_{i+1}), d(u_i), d(h1_i), d(qkv_i), 16 bit): the
        grouped dW launch reads them after the whole group's backward has run, so they cannot share one scratch buffer any more
        (170 MB per layer at 12288 tokens, 2 GB for ViLT-B: nothing on a 288 GB part).  The producing kernels write them directly."""
        if self.dx_c is not None:
            return
        cfg, dev, t16 = eng.cfg, eng.device, eng.t16
        H, Fd, L, M = cfg["hidden"], cfg["ffn"], cfg["layers"], self.M
        pl = (2,) if eng.split else ()          # split mode: (hi, lo) plane pairs
        self.dx_c = [torch.empty(pl + (M, H), dtype=t16, device=dev) for _ in range(L + 1)]
        self.dh_c = [torch.empty(pl + (M, H), dtype=t16, device=dev) for _ in range(L)]
        self.du_l = [torch.empty(pl + (M, Fd), dtype=t16, device=dev) for _ in range(L)]
        self.dqkv_l = [torch.empty(pl + (M, 3 * H), dtype=t16, device=dev) for _ in range(L)]
        self.dz_l = [torch.empty((M, self.r), dtype=t16, device=dev) for _ in range(2 * L)] if self.has_adapters else None      # adapters: d(bottleneck) per site
        # ... and of the LayerNorm backwards' {dgamma, dbeta, bias} partial column sums: reduced by ONE launch per group instead of one per LayerNorm
        lnb = _lib.query("climb_layernorm_bwd_rows_per_block")
        self.part_l = [torch.empty((((M + lnb - 1) // lnb) * 3 * H,), dtype=torch.float32, device=dev) for _ in range(2 * L)]


class HeadState:
    """per-call activations of a task head (fp32, tiny)."""
    pass


class ViltEngine:
    def __init__(self, layout: FlatLayout, device: torch.device, precision: str = "bf16", task_cfgs: Optional[Dict[str, dict]] = None):
        # "fp16": the throughput ("bf16") code path on the IEEE-half build of the library, with a scaled loss gradient
        # "bf16x3" (r06): the fp32 code path with every encoder GEMM on split (hi, lo) bf16 operands, three MFMA products per k-step (csrc/split.hip)
        assert precision in ("fp32", "bf16", "fp16", "bf16x3")
        self.split = precision == "bf16x3"
        self._force_f32 = False         # split mode, inside an adapter's bottleneck: its skinny products (0.1 % of a layer's FLOPs) run on the exact-fp32 GEMM
        self.h16 = None if precision == "fp32" else ("bf16" if self.split else precision)
        if self.h16 is not None:
            _lib.select_h16(self.h16)
            if _lib.h16() != self.h16:
                raise RuntimeError(f"engine precision {precision}: the loaded HIP library computes in {_lib.h16()}")
            self.t16 = _lib.torch_h16()
        self.precision_name = precision
        precision = "fp32" if precision in ("fp32", "bf16x3") else "bf16"      # the two code paths; `h16` says which 16-bit type the second one runs on
        # the GELU pair's epilogue codes (GELU_SAVE above): ws.u holds gelu'(pre-activation) under "deriv", the pre-activation otherwise
        deriv = precision == "bf16" and GELU_SAVE == "deriv"
        self.epi_gelu, self.epi_dgelu = (EPI_GELUD, EPI_MUL) if deriv else (EPI_GELU, EPI_DGELU)
        self.loss_scale = 1.0           # fp16 only: factor on d(logits) of the current backward, divided out of every range in _ready()
        self._grad_dirty = False        # the gradient buffer holds sums of earlier backwards (accumulation without zero_grad)
        # r04: the optimizer in the epilogue of the grouped weight-gradient launch.  `defer_dw` is armed by a caller that promises optimizer.step()
        # is the next reader of the weight gradients (the fused training step); the launch is then held back (`_dw_deferred`) until FusedAdamW.step()
        # runs it with the update in its epilogue.  Anything else that looks at the gradient buffer first calls materialize_dw() (the plain launch).
        self.defer_dw = False
        self._dw_deferred = []          # [(ws, plan)]
        self._grad_extra = False        # the weight matrices' gradient ranges hold something besides zeros (EWC penalty term, an earlier backward)
        self._grad_clean = False        # set by FusedAdamW.step() when it leaves the gradient buffer all zeros; any backward clears it (before_backward)
        self._ewc_fold = None           # r05: {star, fisher, lam, loss}: an EWC term the next FusedAdamW.step() applies inside its passes (park_ewc)
        self._fused_consumed = False    # the last FusedAdamW.step() updated matrices inside the weight-gradient launch (their gradients were never stored)
        self._g16 = None                # data parallel: {stage, scale, ranges}: averaged gradients that still live in the reducer's 16-bit payload buffer
        self._unscale_pending = self._prescaled = False
        self.layout = layout
        self.cfg = layout.cfg
        self.device = torch.device(device)
        self.precision = precision
        self.task_cfgs = task_cfgs or TASK_ARITH
        self.flat: Optional[torch.Tensor] = None        # fp32 master parameters
        self.grad: Optional[torch.Tensor] = None        # fp32 gradients (accumulating, like .grad)
        self._ws: Dict[tuple, Workspace] = {}
        self._shadow = None                             # bf16 copies of the flat buffer (bf16 mode)
        self._shadow_t = None
        self._shadow_version = -1
        self._shadow_stale = False
        self._t_fresh = self._t_updated = None
        self._t_sub = {}
        self._ewc_ws = None
        self._bce_ws = None
        self._head_ws: Dict[tuple, dict] = {}
        self.requires_grad: Dict[str, bool] = {n: True for n in layout.shapes}
        self.grad_ready_hook: Optional[Callable[[int, int], None]] = None   # (lo, hi) flat range whose grads are final
        self._reserve_cell = None        # [reserve] shared with this engine's finalizer (set_cu_reserve)
        self._cu_reserve = 0             # CUs the persistent GEMMs leave free (set_cu_reserve: collectives running under the backward)
        self.touched: List[tuple] = []                  # flat ranges that received gradients in the last backward
        self.saved = None
        self._unused = set()                            # trainable tensors the last forward did not use (their .grad stays None, like torch's)
        self.active_adapter: Optional[str] = None       # name of the adapter applied in forward/backward (None = plain ViLT)
        self.prof = None                                # bench.py: {"kernel": name, "events": [(start, end, flops)]}
        # weight-gradient GEMMs run on a second HIP stream, concurrently with the input-gradient chain they do not feed
        self.overlap_dw = os.environ.get("CLIMB_AMD_OVERLAP_DW", "0") != "0"   # measured slower on MI355X (r01): off
        self._side = None
        self._side_pending = None
        # OPT-IN (CLIMB_AMD_CLS_ONLY_LAST=1): rows of the LAST encoder layer that nothing reads are not computed (DESIGN.md section 5 "CLS rows
        # only").  The only consumer of x_L is the pooler, which reads token 0 of every sequence (REF/modeling/vilt.py:123-124 returns
        # `pooler_output` alone), so after the last layer's attention only the B [CLS] rows go through the out-projection / MLP / final
        # LayerNorm, forward and backward.  Same loss, same gradients (tests/test_gpu_parity.py compares the two steps); off by default so that
        # the default step executes every FLOP of the reference's (bench.py times both).
        self.cls_only_last = os.environ.get("CLIMB_AMD_CLS_ONLY_LAST", "0") != "0" and not self.split      # (the opt-in row pruning has no split path)

    # ------------------------------------------------------------------ buffers
    def allocate(self):
        if self.device.type != "cuda":
            raise RuntimeError("climb_amd.ViltEngine needs a HIP device (cuda:N); there is no CPU path in the product. "
                               "Use oracle/ for CPU checking.")
        _lib.load()
        self.flat = torch.zeros(self.layout.total, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros(self.layout.total, dtype=torch.float32, device=self.device)
        if self.precision == "bf16" or self.split:
            tn_workspace(self.device)
        self._ws.clear()
        self._shadow = None
        self._shadow_version = -1
        self._t_fresh = None
        self._t_updated = None
        self._t_sub = {}

    def view(self, base: torch.Tensor, name: str) -> torch.Tensor:
        o = self.layout.offset[name]
        return base[o:o + self.layout.numel(name)].view(self.layout.shapes[name])

    def p(self, name: str) -> int:
        return self.flat.data_ptr() + 4 * self.layout.offset[name]

    def g(self, name: str) -> int:
        return self.grad.data_ptr() + 4 * self.layout.offset[name]

    def zero_grad(self):
        # (r04) the optimizer step that just ran cleared every gradient it consumed and nothing else had been written: the buffer is zeros already
        if not self._grad_clean:
            self.grad.zero_()
        self._grad_clean = False
        self._g16 = None
        self.touched = []
        self._grad_dirty = False
        self._dw_deferred = []          # gradients nobody asked for are never computed
        if self._ewc_fold is not None:  # (its value was promised to the caller; the gradients are being dropped)
            f, self._ewc_fold = self._ewc_fold, None
            f["loss"].copy_(self.ewc_penalty(f["star"], f["fisher"], f["lam"], add_grad=False))
        self._grad_extra = False
        self._fused_consumed = False

    def materialize_g16(self):
        """Cast averaged gradients the data-parallel reducer left in its 16-bit payload buffer (GradientAllReducer.finish(defer_uncast=True)) back into
        the fp32 gradient buffer: what finish() does itself when nobody promised that FusedAdamW.step() is the next reader."""
        pend, self._g16 = self._g16, None
        if pend:
            for lo, hi in pend["ranges"]:
                _lib.call("climb_uncast_bf16_scale", pend["stage"][lo:hi], self.grad[lo:hi], hi - lo, pend["scale"], _stream())

    def park_ewc(self, star: torch.Tensor, fisher: torch.Tensor, lam: float) -> torch.Tensor:
        """r05: the EWC term of a fused training step whose caller named its optimizer is not written by a pass of its own
        (`ewc_penalty`: 16 B per encoder parameter, then 4 B more when the optimizer re-reads what was parked in the gradient buffer): FusedAdamW.step()
        adds 2 lam F (theta - theta*) to the gradient inside its two passes -- the weight-gradient epilogue and the flat pass, 8 B per parameter -- and
        accumulates the penalty's value.  Returns the tensor that RECEIVES the value when step() runs (REF/.../train_vqa.py:160-170 reads it after the
        step); anything else that looks at the gradients first (`materialize_dw`) writes the term the old way."""
        loss = torch.zeros((), dtype=torch.float32, device=self.device)
        self._ewc_fold = dict(star=star, fisher=fisher, lam=float(lam), loss=loss)
        return loss

    def apply_parked_ewc(self):
        f, self._ewc_fold = self._ewc_fold, None
        if f is not None:
            f["loss"].copy_(self.ewc_penalty(f["star"], f["fisher"], f["lam"], add_grad=True))

    def materialize_dw(self, keep_parked_ewc: bool = False):
        """Run weight-gradient launches that were held back for the optimizer as the plain launches they replace (C += dW)."""
        if not keep_parked_ewc:
            self.apply_parked_ewc()
        held, self._dw_deferred = self._dw_deferred, []
        for ws, plan in held:
            self._launch_dw_plan(plan)
            self._grad_extra = True

    def _launch_dw_plan(self, plan):
        if plan.get("split"):
            self._timed_call("gemm_split_tn", plan["flops"], "climb_gemm_split_tn_grouped", plan["probs"], plan["items"], plan["first"], plan["nwg"], _stream())
        else:
            self._timed_call("gemm_bf16_tn", plan["flops"], "climb_gemm_bf16_tn_grouped", plan["probs"], plan["items"], plan["first"], plan["nwg"], plan["ragged"], _stream())

    def begin_scaled_backward(self, max_dlogit: float):
        """fp16 operands only.  Picks the power-of-two loss scale that puts the largest possible |d(logits)| near 1 (the BCE gradient of a
        64 x 3129 mean is 2.5e-6 per element: below IEEE half's normal range by the time it reaches the first 16-bit GEMM operand) and, if the
        gradient buffer already holds sums of earlier backwards, brings those to the same scale so that _ready() can divide every range once."""
        if self.h16 != "fp16":
            self.loss_scale = 1.0
            return 1.0
        import math
        self.loss_scale = float(2.0 ** max(0, min(24, math.floor(math.log2(128.0 / max(max_dlogit, 1e-30))))))      # |d(logits)| <= 128: 2^9 below half's largest number
        self._prescaled = bool(self._grad_dirty and self.loss_scale != 1.0)
        if self._prescaled:            # exact (a power of two) and undone for the WHOLE buffer by finish_scaled_backward()
            _lib.call("climb_scale_f32", self.grad, self.grad.numel(), self.loss_scale, _stream())
            self._unscale_pending = True
        self._grad_dirty = True
        return self.loss_scale

    def finish_scaled_backward(self):
        """End of a backward on the half build (fused path: after the encoder backward; autograd path: in the last Function's backward):
        divide the loss scale out of the gradient buffer, once.  Ranges this backward did not touch are zero, or were pre-scaled sums."""
        if self._unscale_pending:
            _lib.call("climb_scale_f32", self.grad, self.grad.numel(), 1.0 / self.loss_scale, _stream())
            self._unscale_pending = False

    def is_touched(self, name: str) -> bool:
        o = self.layout.offset[name]
        for lo, hi in self.touched:
            if lo <= o < hi:
                return True
        return False

    def workspace(self, B: int, T: int, gh: int = None, gw: int = None, nseq: int = None, tag: str = None) -> Workspace:
        g0 = self.cfg["image"] // self.cfg["patch"]
        key = (B, T, gh or g0, gw or g0, nseq, tag)
        ws = self._ws.get(key)
        if ws is None:
            if len(self._ws) >= 3:      # bound HBM use when batch shapes vary (last partial batch, replay batches, canvases)
                pending = self.saved["ws"] if self.saved is not None else None
                victim = next(k for k, w in self._ws.items() if w is not pending)      # never the one a pending backward will read
                self._ws.pop(victim)
            ws = self._ws[key] = Workspace(self, B, T, gh, gw, nseq)
        return ws

    # ------------------------------------------------------------------ GEMM dispatch
    # forward:  Y[M,N] = X[M,K] W[N,K]^T ; input grad: dX[M,K] = dY[M,N] W[N,K] ; weight grad: dW[N,K] += dY^T X
    def _gemm_f32(self, A, sam, sak, Bm, sbn, sbk, C, ldc, M, N, K, bias=None, epi=EPI_NONE, aux=None, ldaux=0, aux_out=None, ldauxo=0, beta=0.0,
                  aux2=None, ldaux2=0):
        # split-K (atomic partial sums) only in the throughput mode: the fp32 parity mode stays run-to-run deterministic
        self._timed_call("gemm_f32", 2.0 * M * N * K, "climb_gemm_f32", A, sam, sak, Bm, sbn, sbk, C, ldc, M, N, K, bias, epi, aux, ldaux, aux_out,
                         ldauxo, beta, aux2, ldaux2, 1 if (self.precision == "bf16" or self.split) else 0, _stream())

    def _rank_update(self, dY, lddy, X, ldx, wname, M, N, K):
        """grad(W)[N, K] += dY[M, N]^T X[M, K] for M = batch rows (csrc/heads.hip); exact fp32"""
        if SKINNY:
            self._timed_call("skinny_f32", 2.0 * M * N * K, "climb_rank_update_f32", dY, lddy, X, ldx, self.g(wname), K, M, N, K, _stream())
        else:
            self._gemm_f32(dY, 1, lddy, X, 1, ldx, self.g(wname), K, N, K, M, beta=1.0)

    def _skinny(self, A, lda, B, sbn, sbk, C, ldc, M, N, K, bias=None, epi=0, aux=None, ldaux=0, colsum=None, acol=None):
        """csrc/heads.hip: C = epi(A B^T + bias) for M = batch rows; colsum / acol: parameter-gradient pointers that the column sums of C / of A are
        ADDED to (the bias gradients on either side of the product)."""
        self._timed_call("skinny_f32", 2.0 * M * N * K, "climb_skinny_f32", A, lda, B, sbn, sbk, C, ldc, M, N, K, bias, epi, aux, ldaux, colsum, 1.0,
                         acol, 1.0, _stream())

    # ---- split operands (r06): X / dY are (hi, lo) plane pairs [2][M][.], weights come from the split shadows, results are fp32
    def _split_nt(self, X, ldx, x_lo, Wp, ldw, w_lo, Y, ldy, M, N, K, bias, epi, aux, ldaux):
        self._timed_call("gemm_split_nt", 2.0 * M * N * K, "climb_gemm_split_nt", X, ldx, x_lo, Wp, ldw, w_lo, Y, ldy, M, N, K, bias, epi, aux, ldaux, _stream())

    def split_of(self, src, dst, M, C, mode=0, aux=None):
        """dst (split [2][M][C]) = f(src fp32 [M, C]): 0 copy, 1 gelu, 2 src * gelu'(aux)"""
        _lib.call("climb_split_f32", src, C, dst, C, M * C, M, C, mode, aux, C, _stream())

    def linear_fwd(self, X, wname, bname, Y, M, N, K, epi=EPI_NONE, aux=None, aux_out=None, aux2=None, out_f32=False):
        if self.split and not self._force_f32:
            if epi not in (EPI_NONE, EPI_RESID):
                raise NotImplementedError("bf16x3: fused activation epilogues are not built for split operands")
            self._split_nt(X, K, M * K, self.sp(wname), K, self.layout.total, Y, N, M, N, K, self.p(bname) if bname else None, epi, aux, N)
        elif self.precision == "fp32":
            self._gemm_f32(X, K, 1, self.p(wname), K, 1, Y, N, M, N, K, self.p(bname) if bname else None, epi, aux, N, aux_out, N, 0.0, aux2, N)
        else:
            self._bf16_fwd(X, wname, bname, Y, M, N, K, epi, aux, aux_out, out_f32=out_f32, aux2=aux2)

    def linear_dx(self, dY, wname, dX, M, N, K, epi=EPI_NONE, aux=None):
        """dX[M,K] = dY[M,N] @ W[N,K]   (epi DGELU multiplies by gelu'(aux[M,K]))"""
        if self.split and not self._force_f32:
            if epi not in (EPI_NONE, EPI_RESID):
                raise NotImplementedError("bf16x3: fused activation epilogues are not built for split operands")
            self._split_nt(dY, N, M * N, self.spt(wname), N, self._shadow_t.numel() // 2, dX, K, M, K, N, None, epi, aux, K)
        elif self.precision == "fp32":
            self._gemm_f32(dY, N, 1, self.p(wname), 1, K, dX, K, M, K, N, None, epi, aux, K)
        else:
            self._bf16_dx(dY, wname, dX, M, N, K, epi, aux)

    def linear_dw(self, dY, X, wname, M, N, K, bname=None, ws=None):
        """dW[N,K] += dY[M,N]^T @ X[M,K];  db[N] += colsum(dY) when `bname` is given (fused into the bf16 dW kernel)."""
        want_b = bname is not None and self.requires_grad[bname]
        if not self.requires_grad[wname]:
            if want_b:
                self.bias_grad(dY, self.adt, bname, M, N, ws)
            return
        if self.split and not self._force_f32:          # (shapes the grouped launch does not take: three ordinary weight-gradient launches; the bias gradient rides in two of them)
            self._timed_call("gemm_split_tn", 2.0 * M * N * K, "climb_gemm_split_tn", dY, N, M * N, X, K, M * K, self.g(wname), K, M, N, K,
                             self.g(bname) if want_b else None, _stream())
        elif self.precision == "fp32":
            if want_b:
                self.bias_grad(dY, F32, bname, M, N, ws)
            self._gemm_f32(dY, 1, N, X, 1, K, self.g(wname), K, N, K, M, beta=1.0)
        else:
            self._bf16_dw(dY, X, wname, M, N, K, self.g(bname) if want_b else None)

    # the same three products on a row-STRIDED subset (the last layer's [CLS] rows: row b of an operand is row b * S_pad of a saved [M, .] one)
    def _lin_fwd_ld(self, X, ldx, wname, bname, Y, ldy, M, N, K, epi=EPI_NONE, aux=None, ldaux=0, aux_out=None, ldauxo=0, out_f32=False):
        bias = self.p(bname) if bname else None
        if self.precision == "fp32":
            self._gemm_f32(X, ldx, 1, self.p(wname), K, 1, Y, ldy, M, N, K, bias, epi, aux, ldaux, aux_out, ldauxo)
        else:
            self._timed_call("gemm_bf16_nt_rows", 2.0 * M * N * K, "climb_gemm_bf16_nt", X, ldx, self.sp(wname), K, Y, ldy, F32 if out_f32 else BF16, M, N, K,
                             bias, epi, aux, ldaux, aux_out, ldauxo, None, 0, _stream())

    def _lin_dx_ld(self, dY, lddy, wname, dX, lddx, M, N, K, epi=EPI_NONE, aux=None, ldaux=0):
        if self.precision == "fp32":
            self._gemm_f32(dY, lddy, 1, self.p(wname), 1, K, dX, lddx, M, K, N, None, epi, aux, ldaux)
        else:
            self._timed_call("gemm_bf16_nt_rows", 2.0 * M * N * K, "climb_gemm_bf16_nt", dY, lddy, self.spt(wname), N, dX, lddx, BF16, M, K, N, None, epi, aux,
                             ldaux, None, 0, None, 0, _stream())

    def _lin_dw_ld(self, dY, lddy, X, ldx, wname, M, N, K, bname=None, ws=None):
        want_b = bname is not None and self.requires_grad[bname]
        if want_b and (self.precision == "fp32" or not self.requires_grad[wname]):
            csr = _lib.query("climb_colsum_rows_per_block")
            _lib.call("climb_colsum", dY, lddy, self.adt, None, 0, ws.part, M, N, _stream())
            _lib.call("climb_colreduce", ws.part, N, (M + csr - 1) // csr, self.g(bname), N, 1.0, _stream())
        if not self.requires_grad[wname]:
            return
        if self.precision == "fp32":
            self._gemm_f32(dY, 1, lddy, X, 1, ldx, self.g(wname), K, N, K, M, beta=1.0)
        else:
            self._timed_call("gemm_bf16_tn", 2.0 * M * N * K, "climb_gemm_bf16_tn", dY, lddy, X, ldx, self.g(wname), K, M, N, K,
                             self.g(bname) if want_b else None, _stream())

    def reduce3(self, part, nblk, ncols, n0, n1, n2):
        """{dgamma, dbeta, colsum} partials -> three parameter gradients, one launch (names may be None / frozen)."""
        rg = self.requires_grad
        ptrs = [self.g(n) if (n is not None and rg[n]) else None for n in (n0, n1, n2)]
        _lib.call("climb_colreduce3", part, 3 * ncols, nblk, ptrs[0], ptrs[1], ptrs[2], ncols, 1.0, _stream())

    def dw_async(self, *args, **kw):
        """linear_dw on the side stream: dW needs dY and the saved input, and nothing on the dX chain needs dW, so the two
        GEMM families of a layer overlap (this also fills the tail waves the 768-wide GEMMs leave on 256 CUs)."""
        if not self.overlap_dw:
            return self.linear_dw(*args, **kw)
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        ev = torch.cuda.Event()
        ev.record(main)
        self._side.wait_event(ev)
        with torch.cuda.stream(self._side):
            self.linear_dw(*args, **kw)
            done = torch.cuda.Event()
            done.record(self._side)
        self._side_pending = done

    def join_side(self):
        """Main stream waits for every outstanding weight-gradient GEMM (called before a buffer they read is overwritten
        and before a layer's gradients are reported final)."""
        if self._side_pending is not None:
            torch.cuda.current_stream().wait_event(self._side_pending)
            self._side_pending = None

    def bias_grad_from_part(self, part_ptr, stride, nblk, bname, ncols):
        if bname is not None and self.requires_grad[bname]:
            _lib.call("climb_colreduce", part_ptr, stride, nblk, self.g(bname), ncols, 1.0, _stream())

    def bias_grad(self, dY, dtype, bname, M, C, ws):
        """db += colsum(dY)"""
        if not self.requires_grad[bname]:
            return
        csr = _lib.query("climb_colsum_rows_per_block")
        if self.split and dY.dim() == 3:          # a split operand: the column sums of its hi plane + those of its lo plane
            for plane in (dY[0], dY[1]):
                _lib.call("climb_colsum", plane, C, BF16, None, 0, ws.part, M, C, _stream())
                _lib.call("climb_colreduce", ws.part, C, (M + csr - 1) // csr, self.g(bname), C, 1.0, _stream())
            return
        _lib.call("climb_colsum", dY, C, dtype, None, 0, ws.part, M, C, _stream())
        _lib.call("climb_colreduce", ws.part, C, (M + csr - 1) // csr, self.g(bname), C, 1.0, _stream())

    # ------------------------------------------------------------------ bf16 operand shadows + GEMM dispatch
    def _linear_weight_names(self):
        """2-D encoder weights that appear as the B operand of an input-gradient GEMM (need a [K,N] transposed shadow)."""
        out = []
        for i in range(self.cfg["layers"]):
            l = f"{ENC}encoder.layer.{i}."
            H, Fd = self.cfg["hidden"], self.cfg["ffn"]
            out += [(l + "attention.attention.query.weight", 3 * H, H), (l + "attention.output.dense.weight", H, H),
                    (l + "intermediate.dense.weight", Fd, H), (l + "output.dense.weight", H, Fd)]
            for t, r in self.layout.adapters.items():
                for site in ("attention.output", "output"):
                    a = f"{l}{site}.adapters.{t}."
                    out += [(a + "adapter_down.0.weight", r, H), (a + "adapter_up.weight", H, r)]
        return out

    def _build_shadow(self):
        import numpy as np
        dev = self.device
        npl = 2 if self.split else 1          # split mode: [2][total] -- the hi plane, then the lo plane
        self._shadow = torch.empty(npl * self.layout.total, dtype=self.t16, device=dev)
        rows, off = [], 0
        self._t_off = {}
        for name, N, K in self._linear_weight_names():
            self._t_off[name] = off
            rows.append((self.layout.offset[name], off, N, K))
            off += N * K
        self._shadow_t = torch.empty(npl * off, dtype=self.t16, device=dev)
        self._t_table = torch.from_numpy(np.array(rows, dtype=np.int64)).to(dev)
        self._t_n = len(rows)

    def refresh_shadow(self, cast: bool = True):
        """bf16 copies of the weights for the MFMA operands: refreshed when the fp32 master changed (torch in-place ops
        bump the version counter; our fused AdamW refreshes the straight shadow itself and calls params_updated)."""
        if self.precision != "bf16" and not self.split:
            return
        if self._shadow is None:
            self._build_shadow()
            self._shadow_version = -1
        ver = self.flat._version
        if ver == self._shadow_version and not self._shadow_stale:
            return
        st = _stream()
        table, tn = self._t_table, self._t_n
        if self.split:
            tot, tt = self.layout.total, self._shadow_t.numel() // 2
            if self._shadow_stale == "transpose-only" and self._t_updated is not None:
                # the optimizer ran in the weight-gradient epilogue and wrote the planes of `_t_fresh` there: what is left are the GEMM weights the FLAT pass
                # updated (it writes no planes) -- the patch projection, a problem with stream-K tiles -- re-split one by one, and their transposes
                import numpy as np
                fresh, upd = frozenset(self._t_fresh or ()), frozenset(self._t_updated)
                key = ("split", fresh, upd)
                sub = self._t_sub.get(key)
                if sub is None:
                    todo = [(name, N, K) for name, N, K in self._linear_weight_names() if name not in fresh and any(c in upd for c in self._covered(name, N * K))]
                    pw = ENC + "embeddings.patch_embeddings.projection.weight"
                    flat_only = [pw] if (pw in upd and pw not in fresh) else []
                    rows = np.array([(self.layout.offset[name], self._t_off[name], N, K) for name, N, K in todo], dtype=np.int64).reshape(-1, 4)
                    if len(self._t_sub) >= 16:
                        self._t_sub.pop(next(iter(self._t_sub)))
                    sub = self._t_sub[key] = (torch.from_numpy(rows).to(self.device), len(todo), [(self.layout.offset[n], N * K) for n, N, K in todo] +
                                              [(self.layout.offset[n], self.layout.numel(n)) for n in flat_only])
                stab, sn, spans = sub
                for o, n in spans:
                    _lib.call("climb_split_f32", self.flat[o:o + n], n, self._shadow[o:], n, tot, 1, n, 0, None, 0, st)
                if sn:
                    _lib.call("climb_transpose_bf16_batched", self._shadow, self._shadow_t, stab, sn, 96, st)
                    _lib.call("climb_transpose_bf16_batched", self._shadow[tot:], self._shadow_t[tt:], stab, sn, 96, st)
            else:          # both planes of the straight shadow in one pass over the master weights, then the two planes' transposes
                _lib.call("climb_split_f32", self.flat, tot, self._shadow, tot, tot, 1, tot, 0, None, 0, st)
                _lib.call("climb_transpose_bf16_batched", self._shadow, self._shadow_t, table, tn, 96, st)
                _lib.call("climb_transpose_bf16_batched", self._shadow[tot:], self._shadow_t[tt:], table, tn, 96, st)
            self._t_fresh = None
            self._t_updated = None
            self._shadow_version = ver
            self._shadow_stale = False
            return
        if self._shadow_stale != "transpose-only":
            _lib.call("climb_cast_bf16", self.flat, self._shadow, self.layout.total, st)
        elif self._t_fresh or self._t_updated is not None:
            # the optimizer ran in the weight-gradient epilogue for `_t_fresh` and wrote their transposed shadows there; `_t_updated` (r06) = the tensors
            # the optimizer touched AT ALL this step: a matrix it did not update (a frozen base under adapters: 85 M elements, 0.1 ms per step) keeps
            # the transposed shadow it has
            fresh = frozenset(self._t_fresh or ())
            upd = None if self._t_updated is None else frozenset(self._t_updated)
            key = (fresh, upd)
            sub = self._t_sub.get(key)
            if sub is None:
                import numpy as np
                rows = [(self.layout.offset[name], self._t_off[name], N, K) for name, N, K in self._linear_weight_names()
                        if name not in fresh and (upd is None or any(c in upd for c in self._covered(name, N * K)))]      # (a fused QKV row is named by its query weight)
                if len(self._t_sub) >= 16:
                    self._t_sub.pop(next(iter(self._t_sub)))
                sub = self._t_sub[key] = (torch.from_numpy(np.array(rows, dtype=np.int64).reshape(-1, 4)).to(self.device), len(rows))
            table, tn = sub
        if tn:
            _lib.call("climb_transpose_bf16_batched", self._shadow, self._shadow_t, table, tn, 96, st)
        self._t_fresh = None
        self._t_updated = None
        self._shadow_version = ver
        self._shadow_stale = False

    def shadow_ptr(self):
        """bf16 weight shadow the fused AdamW refreshes in the same pass (None in fp32 mode)."""
        if self.precision != "bf16":          # (split mode: the optimizer leaves the planes to refresh_shadow())
            return None
        if self._shadow is None:
            self.refresh_shadow()          # first use: full cast, so tensors the optimiser skips have valid shadows too
        return self._shadow

    def params_updated(self, shadow_fresh: bool = False, t_fresh=None, updated=None):
        """`updated` (names, optional): everything the optimizer changed this step -- anything else keeps its transposed shadow.  Two updates without a
        refresh in between (nobody ran a forward) accumulate."""
        prev_pending = self._shadow_stale == "transpose-only"
        prev_upd = getattr(self, "_t_updated", None)
        self._shadow_stale = "transpose-only" if (shadow_fresh and self._shadow_stale in (False, "transpose-only")) else True
        self._t_fresh = set(t_fresh) if (shadow_fresh and t_fresh and not prev_pending) else None
        if shadow_fresh and updated is not None and (not prev_pending or prev_upd is not None):
            self._t_updated = set(updated) | (set(prev_upd) if (prev_pending and prev_upd is not None) else set())
        else:
            self._t_updated = None

    def sp(self, name: str) -> int:
        return self._shadow.data_ptr() + 2 * self.layout.offset[name]

    def spt(self, name: str) -> int:
        return self._shadow_t.data_ptr() + 2 * self._t_off[name]

    def _timed_call(self, kernel, flops, name, *args):
        prof = self.prof
        if prof is not None and prof["kernel"] == kernel:
            pool = prof.get("pool")            # pre-created events: creating hundreds inside a timed region costs milliseconds now and then
            if pool:
                e0, e1 = pool.pop(), pool.pop()
            else:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.call(name, *args)
            e1.record()
            # (bench.py's per-kind table) an NT GEMM launch is named by its output width, reduction depth, epilogue and output type
            kind = f"N{args[8]}_K{args[9]}_epi{args[11]}_{'f32' if args[6] == F32 else 'h16'}" if name == "climb_gemm_bf16_nt" else name
            if name == "climb_gemm_split_nt":
                kind = f"N{args[9]}_K{args[10]}_epi{args[12]}_split"
            if name == "climb_gemm_split_nt_act":
                kind = f"N{args[12]}_K{args[13]}_epi{args[15]}_split"
            prof["events"].append((e0, e1, flops, kind))
        else:
            _lib.call(name, *args)

    def _bf16_fwd(self, X, wname, bname, Y, M, N, K, epi=EPI_NONE, aux=None, aux_out=None, out_f32=False, aux2=None):
        self._timed_call("gemm_bf16_nt", 2.0 * M * N * K, "climb_gemm_bf16_nt", X, K, self.sp(wname), K, Y, N, F32 if out_f32 else BF16, M, N, K,
                         self.p(bname) if bname else None, epi, aux, N, aux_out, N, aux2, N, _stream())

    def _bf16_dx(self, dY, wname, dX, M, N, K, epi=EPI_NONE, aux=None):
        # dX[M,K] = dY[M,N] W[N,K] = dY (W^T)^T : NT GEMM against the transposed shadow [K,N]
        self._timed_call("gemm_bf16_nt", 2.0 * M * N * K, "climb_gemm_bf16_nt", dY, N, self.spt(wname), N, dX, K, BF16, M, K, N, None, epi, aux, K,
                         None, 0, None, 0, _stream())

    def _bf16_dw(self, dY, X, wname, M, N, K, dbias=None):
        self._timed_call("gemm_bf16_tn", 2.0 * M * N * K, "climb_gemm_bf16_tn", dY, N, X, K, self.g(wname), K, M, N, K, dbias, _stream())

    @property
    def adt(self):
        return F32 if self.precision == "fp32" else BF16

    @property
    def odt(self):
        """dtype code of a tensor that only GEMMs read (LayerNorm outputs, the LayerNorm backward's cast)"""
        return SPLIT if self.split else self.adt

    # ------------------------------------------------------------------ encoder forward
    def encoder_forward(self, input_ids, token_type_ids, attention_mask, pixel_values, img_type: torch.Tensor, save: bool = True,
                        pixel_mask: Optional[torch.Tensor] = None, inputs_embeds: Optional[torch.Tensor] = None):
        """[B,T] int64 ids/types/mask, [B,3,Hc,Wc] fp32 pixels, img_type int32 [B] (HF `image_token_type_idx` per sequence).
        `pixel_mask` None = every image fills the 384x384 canvas (benchmark shape); int64 [B,Hc,Wc] = padded variable-resolution
        batch as `ViltProcessor` produces it (HF:92-178 path).  Returns pooled [B,H] fp32 (HF:636-663 pooler_output)."""
        cfg, L = self.cfg, self.layout
        B, T = token_type_ids.shape
        if inputs_embeds is not None:      # [B, >= T, H] fp32 in place of the word-embedding lookup (ViLT-BERT); no gradient flows into it
            assert inputs_embeds.dtype == torch.float32 and inputs_embeds.is_contiguous() and inputs_embeds.shape[0] == B and inputs_embeds.shape[1] >= T
        H, Fd, nh = cfg["hidden"], cfg["ffn"], cfg["heads"]
        P_ = cfg["patch"]
        Hc, Wc = int(pixel_values.shape[-2]), int(pixel_values.shape[-1])
        g0 = cfg["image"] // P_
        if pixel_values.shape[1] != cfg["channels"] or Hc % P_ or Wc % P_:
            raise ValueError(f"pixels must be [B,{cfg['channels']},H,W] with H, W multiples of {P_}; got {tuple(pixel_values.shape)}")
        if pixel_mask is None and (Hc, Wc) != (cfg["image"], cfg["image"]):
            raise ValueError("a canvas other than 384x384 needs its pixel_mask (variable-resolution path)")
        gh, gw = Hc // P_, Wc // P_
        var = pixel_mask is not None
        nseq = None
        if T + 1 + gh * gw > 288:
            # A padded batch that mixes portrait and landscape images: the canvas is up to 20 x 20 patches, yet no image has more
            # than 12 x 20 valid ones (processor rule).  Pack every sample's valid patches (HF:136-159 keeps max_b(h_b*w_b) rows too).
            nseq = getattr(pixel_mask, "_climb_max_patches", None) if var else None     # set by DeviceImagePipeline: no host sync
            if var and nseq is None:
                hv = (pixel_mask[:, ::P_, 0] != 0).sum(1)
                wv = (pixel_mask[:, 0, ::P_] != 0).sum(1)
                nseq = int((hv * wv).max().item())
            if nseq is None or T + 1 + nseq > 288:
                raise NotImplementedError(f"sequence of {T + 1 + (nseq or gh * gw)} tokens exceeds the 288 this build sizes its attention tiles for")
        ws = self.workspace(B, T, gh, gw, nseq)
        if self._dw_deferred and any(w is ws for w, _ in self._dw_deferred):
            # a held-back weight-gradient launch still points at this workspace's saved activations (a no-grad forward between backward and step,
            # or a step that never came): run it now, as the plain launch, before they are overwritten
            self.materialize_dw()
        if not save and self.saved is not None and self.saved["ws"] is ws:
            # a grad-enabled forward of this shape is still waiting for its backward (reference-style autograd path: an evaluation or a
            # teacher pass between `model(...)` and `loss.backward()`): its saved activations live in `ws`, so this pass gets its own
            ws = self.workspace(B, T, gh, gw, nseq, tag="nograd")
        st = _stream()
        adt = self.adt
        self.refresh_shadow()
        e = ENC + "embeddings."
        ws.img_type.copy_(img_type)
        _lib.call("climb_key_bias", attention_mask, ws.key_bias, B, T, ws.S, ws.S_pad, st)
        x0 = ws.x[0]
        _lib.call("climb_embed_text_fwd", input_ids if inputs_embeds is None else None, token_type_ids,
                  self.p(e + "text_embeddings.word_embeddings.weight") if inputs_embeds is None else inputs_embeds,
                  self.p(e + "text_embeddings.token_type_embeddings.weight"), self.p(e + "text_embeddings.position_embeddings.weight"),
                  self.p(e + "text_embeddings.LayerNorm.weight"), self.p(e + "text_embeddings.LayerNorm.bias"),
                  self.p(e + "token_type_embeddings.weight"), cfg["ln_eps"], x0, B, T, ws.S_pad, H, ws.tmean, ws.trstd,
                  0 if inputs_embeds is None else int(inputs_embeds.shape[1]), st)
        _lib.call("climb_im2col", pixel_values, ws.a_patch, adt, B, cfg["channels"], Hc, Wc, P_, st)
        if var:
            _lib.call("climb_patch_grid_dims", pixel_mask, B, Hc, Wc, P_, ws.dims, st)
        Kp = cfg["channels"] * cfg["patch"] ** 2
        a_patch = ws.a_patch
        if self.split:
            self.split_of(ws.a_patch, ws.a_patch_s, B * ws.NP, Kp)
            a_patch = ws.a_patch_s
        self.linear_fwd_f32out(a_patch, e + "patch_embeddings.projection.weight", e + "patch_embeddings.projection.bias", ws.proj,
                               B * ws.NP, H, Kp)
        _lib.call("climb_assemble_image", ws.proj, self.p(e + "cls_token"), self.p(e + "position_embeddings"),
                  self.p(e + "token_type_embeddings.weight"), ws.img_type, ws.dims if var else None, x0, ws.key_bias, B, T, ws.NP, gw, g0,
                  ws.S_pad, H, 1 if ws.compact else 0, st)
        M = ws.M
        ad = self.active_adapter
        r = self.layout.adapters[ad] if ad is not None else 0
        prune = self.cls_only_last and ad is None
        odt = self.odt
        for i in range(cfg["layers"]):
            l = f"{ENC}encoder.layer.{i}."
            x = ws.x[i]
            _lib.call("climb_layernorm_fwd", x, H, self.p(l + "layernorm_before.weight"), self.p(l + "layernorm_before.bias"), cfg["ln_eps"],
                      ws.xn[i], H, odt, ws.mean1[i], ws.rstd1[i], M, H, st)
            # fused QKV projection: q/k/v weights are adjacent in the flat buffer (HF:325-327 as one [2304,768] GEMM)
            self.linear_fwd(ws.xn[i], l + "attention.attention.query.weight", l + "attention.attention.query.bias", ws.qkv[i], M, 3 * H, H)
            if self.split:          # three-product bf16 MFMAs on planes formed inside the kernel; ctx as fp32 (the backward's softmax row term) AND as the out-projection's operand
                _lib.call("climb_attn_fwd_split", ws.qkv[i], ws.key_bias, ws.ctx[i], ws.ctx_s[i], M * H, ws.lse[i], B, ws.S_pad, nh, cfg["head_dim"], st)
            else:
                self.attn_fwd(ws.qkv[i], ws.key_bias, ws.ctx[i], ws.lse[i], B, ws.S_pad)
            if prune and i == cfg["layers"] - 1:
                # last layer: only the [CLS] row of every sequence is read downstream (row b * S_pad of the [M, .] operands)
                SH = ws.S_pad * H
                self._lin_fwd_ld(ws.ctx[i], SH, l + "attention.output.dense.weight", l + "attention.output.dense.bias", ws.h1c, H, B, H, H,
                                 EPI_RESID, x, SH, out_f32=True)
                _lib.call("climb_layernorm_fwd", ws.h1c, H, self.p(l + "layernorm_after.weight"), self.p(l + "layernorm_after.bias"), cfg["ln_eps"],
                          ws.hnc, H, adt, ws.mean2c, ws.rstd2c, B, H, st)
                self._lin_fwd_ld(ws.hnc, H, l + "intermediate.dense.weight", l + "intermediate.dense.bias", ws.ac, Fd, B, Fd, H, self.epi_gelu,
                                 None, 0, ws.uc, Fd)
                self._lin_fwd_ld(ws.ac, Fd, l + "output.dense.weight", l + "output.dense.bias", ws.xLc, H, B, H, Fd, EPI_RESID, ws.h1c, H, out_f32=True)
                continue
            if self.split and ad is None:
                self.linear_fwd_resid(ws.ctx_s[i], l + "attention.output.dense.weight", l + "attention.output.dense.bias", ws.h1[i], M, H, H, x)
            elif self.split:          # h1 = x + y + up(silu(down(y))), y = Wo ctx + bo in fp32
                a_ = f"{l}attention.output.adapters.{ad}."
                self.linear_fwd(ws.ctx_s[i], l + "attention.output.dense.weight", l + "attention.output.dense.bias", ws.ya[i], M, H, H)
                self.adapter_fwd(a_, ws.ya[i], x, ws.za[i], ws.sa[i], ws.h1[i], M, H, r)
            elif ad is None:
                self.linear_fwd_resid(ws.ctx[i], l + "attention.output.dense.weight", l + "attention.output.dense.bias", ws.h1[i], M, H, H, x)
            else:   # h1 = x + y + up(silu(down(y))),  y = Wo ctx + bo
                a_ = f"{l}attention.output.adapters.{ad}."
                self.linear_fwd(ws.ctx[i], l + "attention.output.dense.weight", l + "attention.output.dense.bias", ws.ya[i], M, H, H)
                self.adapter_fwd(a_, ws.ya[i], x, ws.za[i], ws.sa[i], ws.h1[i], M, H, r)
            _lib.call("climb_layernorm_fwd", ws.h1[i], H, self.p(l + "layernorm_after.weight"), self.p(l + "layernorm_after.bias"), cfg["ln_eps"],
                      ws.hn[i], H, odt, ws.mean2[i], ws.rstd2[i], M, H, st)
            if self.split and SPLIT_FUSED_ACT and _lib.query_arg("climb_gemm_split_nt_takes_act", M, Fd, H):
                wn = l + "intermediate.dense.weight"
                self._timed_call("gemm_split_nt", 2.0 * M * Fd * H, "climb_gemm_split_nt_act", ws.hn[i], H, M * H, self.sp(wn), H, self.layout.total, ws.u[i], Fd, ws.a[i], Fd, M * Fd,
                                 M, Fd, H, self.p(l + "intermediate.dense.bias"), EPI_GELU_SP, None, 0, _stream())
            elif self.split:          # u = W1 hn + b1 in fp32 (the backward evaluates gelu' on it), a = gelu(u) as the down-projection's operand
                self.linear_fwd(ws.hn[i], l + "intermediate.dense.weight", l + "intermediate.dense.bias", ws.u[i], M, Fd, H)
                self.split_of(ws.u[i], ws.a[i], M, Fd, mode=1)
            else:
                self.linear_fwd(ws.hn[i], l + "intermediate.dense.weight", l + "intermediate.dense.bias", ws.a[i], M, Fd, H, self.epi_gelu, None, ws.u[i])
            if ad is None:
                self.linear_fwd_resid(ws.a[i], l + "output.dense.weight", l + "output.dense.bias", ws.x[i + 1], M, H, Fd, ws.h1[i])
            else:
                a_ = f"{l}output.adapters.{ad}."
                self.linear_fwd(ws.a[i], l + "output.dense.weight", l + "output.dense.bias", ws.yo[i], M, H, Fd)
                self.adapter_fwd(a_, ws.yo[i], ws.h1[i], ws.zo[i], ws.so[i], ws.x[i + 1], M, H, r)
        xL, ldxL = (ws.xLc, H) if prune else (ws.x[cfg["layers"]], ws.S_pad * H)
        # final LayerNorm only on the row the pooler consumes (token 0 = text [CLS]); `last_hidden_state` is never
        # used by CLiMB (REF/modeling/vilt.py:123-124), so the other S-1 rows are dead work we skip
        _lib.call("climb_layernorm_fwd", xL, ldxL, self.p(ENC + "layernorm.weight"), self.p(ENC + "layernorm.bias"), cfg["ln_eps"],
                  ws.clsn, H, F32, ws.fmean, ws.frstd, B, H, st)
        if SKINNY:                        # r04: every row of a 16-column strip in one workgroup, K split inside it: tanh in the epilogue, no atomics
            self._skinny(ws.clsn, H, self.p(ENC + "pooler.dense.weight"), H, 1, ws.pooled, H, B, H, H, self.p(ENC + "pooler.dense.bias"), epi=1)
        elif self.precision == "bf16":    # skinny GEMM (M = batch): split-K over all CUs, then the activation (61 -> ~20 us at bs = 64)
            self._gemm_f32(ws.clsn, H, 1, self.p(ENC + "pooler.dense.weight"), H, 1, ws.pooled, H, B, H, H, self.p(ENC + "pooler.dense.bias"))
            _lib.call("climb_elementwise", 6, ws.pooled, None, ws.pooled, B * H, 1.0, st)
        else:
            self._gemm_f32(ws.clsn, H, 1, self.p(ENC + "pooler.dense.weight"), H, 1, ws.pooled, H, B, H, H, self.p(ENC + "pooler.dense.bias"), EPI_TANH)
        # the word-embedding table is bypassed by inputs_embeds: it receives no gradient and the optimizer must skip it (torch: grad None)
        self._unused = {e + "text_embeddings.word_embeddings.weight"} if inputs_embeds is not None else set()
        if save:
            self._generation = getattr(self, "_generation", 0) + 1
            # ViLT-BERT: `inputs_embeds` is BertParams' reusable output buffer, which the next BERT forward of this shape overwrites -- the
            # embedding backward reads it, so the saved copy is its own tensor (7.9 MB at bs = 64)
            self.saved = dict(ws=ws, input_ids=input_ids, token_type_ids=token_type_ids, adapter=ad, var=var, generation=self._generation, cls_only=prune,
                              inputs_embeds=inputs_embeds.clone() if inputs_embeds is not None else None)
            self.last_ws = ws
        return ws.pooled

    def linear_fwd_f32out(self, X, wname, bname, Y, M, N, K):
        if self.precision == "fp32":          # (and split: linear_fwd dispatches)
            self.linear_fwd(X, wname, bname, Y, M, N, K)
        else:
            self._bf16_fwd(X, wname, bname, Y, M, N, K, EPI_NONE, None, None, out_f32=True)

    def linear_fwd_resid(self, X, wname, bname, Y, M, N, K, resid):
        if self.precision == "fp32":          # (and split)
            self.linear_fwd(X, wname, bname, Y, M, N, K, EPI_RESID, resid)
        else:
            self._bf16_fwd(X, wname, bname, Y, M, N, K, EPI_RESID, resid, None, out_f32=True)

    def attn_fwd(self, qkv, key_bias, ctx, lse, B, S_pad):
        cfg = self.cfg
        if self.precision == "fp32":
            _lib.call("climb_attn_fwd_f32", qkv, key_bias, ctx, lse, B, S_pad, cfg["heads"], cfg["head_dim"], _stream())
        else:
            _lib.call("climb_attn_fwd_bf16", qkv, key_bias, ctx, lse, B, S_pad, cfg["heads"], cfg["head_dim"], _stream())

    def attn_bwd(self, qkv, key_bias, dctx, ctx, lse, delta, dqkv, B, S_pad):
        cfg = self.cfg
        st = _stream()
        if self.precision == "fp32":
            _lib.call("climb_attn_delta", dctx, ctx, self.adt, delta, B, S_pad, cfg["heads"], st)
            _lib.call("climb_attn_bwd_f32", qkv, key_bias, dctx, lse, delta, dqkv, B, S_pad, cfg["heads"], cfg["head_dim"], st)
        else:   # the bf16 kernel computes delta in its first phase (`delta` is its scratch)
            _lib.call("climb_attn_bwd_bf16", qkv, key_bias, dctx, ctx, lse, delta, dqkv, B, S_pad, cfg["heads"], cfg["head_dim"], st)

    # ------------------------------------------------------------------ deferred, grouped weight gradients
    def _dw_group_size(self, ws: Workspace, ad) -> int:
        """Layers per grouped weight-gradient launch for this backward; 0 = the immediate per-GEMM path."""
        if (self.precision != "bf16" and not self.split) or (ws.M % 128) or self.overlap_dw:
            return 0
        if ad is not None:
            # under an active adapter the base is normally frozen (train_adapter); the backward then keeps one d(y) scratch for all layers,
            # which a DEFERRED gradient of a trainable base weight would read too late: that combination takes the immediate path
            l0 = f"{ENC}encoder.layer."
            if any(self.requires_grad[f"{l0}{i}.{n}"] for i in range(self.cfg["layers"]) for n in ("output.dense.weight", "attention.output.dense.weight")):
                return 0
        if _DW_GROUP is not None:
            return max(0, int(_DW_GROUP))
        return self._ready_group() if self.grad_ready_hook is not None else self.cfg["layers"]

    def _ready_group(self) -> int:
        """Layers per reported gradient range.  Under a data-parallel hook this must NOT depend on the local batch (ranks with different shard
        sizes take different weight-gradient paths -- e.g. a last batch of 2 + 1 examples -- but have to issue the same collectives): always
        groups of 4 layers (or CLIMB_AMD_DW_GROUP) from the top down, whichever path computed them."""
        if self.grad_ready_hook is None:
            return 1
        if _DW_GROUP is not None and int(_DW_GROUP) > 0:
            return int(_DW_GROUP)
        # r05: a reducer that DEFERS its collectives to after the backward (GradientAllReducer.overlap False: the setting bench.py's warm-up trial, or
        # CLIMB_AMD_DP_OVERLAP=0, chose for every rank alike) gains nothing from gradients that become final in chunks: one group = one grouped
        # weight-gradient launch (three launches cost 0.3 ms more than one: three stream-K tails) and one collective
        owner = getattr(self.grad_ready_hook, "__self__", None)
        if owner is not None and getattr(owner, "overlap", True) is False:
            return self.cfg["layers"]
        return 4

    def set_cu_reserve(self, n: int):
        """Leave `n` CUs to somebody else (RCCL's collectives under the backward: parallel.GradientAllReducer.reserve_cus): the persistent GEMMs --
        one workgroup per CU, walking the tiles -- are launched on the remaining ones until this is called again with 0."""
        n = max(0, int(n))
        if n == self._cu_reserve:
            return
        if self.precision == "bf16":
            # the persistent NT grid is a library-wide setting: remember what was in force before the FIRST reserve (the library default, or a grid
            # pinned through CLIMB_AMD_OPTIONS / tools) and put exactly that back with n = 0
            #  the saved value lives at module level -- the setting is the library's, not this engine's: engines that reserve in turn must
            # not restore each other's reduced grids -- and 0 ("uncapped") counts as every CU when the reduced grid is computed
            global _NT_GRID_BEFORE_RESERVE, _NT_RESERVING
            if not len(_NT_RESERVING):
                _NT_GRID_BEFORE_RESERVE = int(_lib.query_arg("climb_get_option", 9))
            base = _NT_GRID_BEFORE_RESERVE
            if n == 0:
                _NT_RESERVING.discard(self)
                if not len(_NT_RESERVING):
                    _lib.call("climb_set_option", 9, base)
            else:
                if self not in _NT_RESERVING:
                    _NT_RESERVING.add(self)
                    if self._reserve_cell is None:
                        self._reserve_cell = [0]
                        weakref.finalize(self, _restore_nt_grid_if_unreserved, self._reserve_cell)
                full = base if base > 0 else torch.cuda.get_device_properties(self.device).multi_processor_count
                _lib.call("climb_set_option", 9, max(8, (full - n) // 8 * 8))
        self._cu_reserve = n
        if self._reserve_cell is not None:
            self._reserve_cell[0] = n

    def _dw_defer(self, pending: list, dY, X, wname, M, N, K, bname=None, ws=None):
        """linear_dw, but recorded for the group's launch when the shape fits its 256 x 256 tiles (else run now)."""
        want_b = bname is not None and self.requires_grad[bname]
        if not self.requires_grad[wname] or (M % 128) or (N % 8) or (K % 8):
            return self.linear_dw(dY, X, wname, M, N, K, bname, ws)
        pending.append((dY, X, wname, bname if want_b else None, M, N, K))

    def _dw_flush(self, ws: Workspace, pending: list):
        if not pending:
            return
        import numpy as np
        sp = self.split
        if sp and any((N % 256) or (K % 256) for _, _, _, _, M, N, K in pending):          # the split launch takes whole tiles only: run those now
            odd = [p for p in pending if (p[5] % 256) or (p[6] % 256)]
            pending[:] = [p for p in pending if not ((p[5] % 256) or (p[6] % 256))]
            for dY, X, w, b, M, N, K in odd:
                self.linear_dw(dY, X, w, M, N, K, b, ws)
            if not pending:
                return
        #  the planner reads option 22 (staggered epilogues) when the plan is BUILT: its value is part of the key, so an in-process A/B re-plans
        key = (self._cu_reserve, int(_lib.query_arg("climb_get_option", 22))) + tuple((w, b, dY.data_ptr(), X.data_ptr()) for dY, X, w, b, M, N, K in pending)
        plan = ws.dw_plans.get(key)
        if plan is None:
            nwg = max(8, (torch.cuda.get_device_properties(self.device).multi_processor_count - self._cu_reserve) // 8 * 8)
            rec = np.zeros(len(pending), dtype=[("A", "<u8"), ("B", "<u8"), ("C", "<u8"), ("dbias", "<u8"), ("lda", "<i8"), ("ldb", "<i8"), ("ldc", "<i8"),
                                                ("M", "<i4"), ("N", "<i4"), ("K", "<i4"), ("reserved", "<i4")])
            assert rec.dtype.itemsize == 72
            for r, (dY, X, w, b, M, N, K) in zip(rec, pending):
                r["A"], r["B"], r["C"], r["dbias"] = dY.data_ptr(), X.data_ptr(), self.g(w), (self.g(b) if b is not None else 0)
                r["lda"], r["ldb"], r["ldc"], r["M"], r["N"], r["K"] = N, K, K, M, N, K
                if sp:          # three phases over the (hi, lo) planes stacked along the tokens (csrc/gemm_bf16_tnp.hip, SPLIT)
                    r["M"], r["reserved"] = 3 * M, M // 64
            Ms, Ns, Ks = (np.ascontiguousarray(rec[f], dtype=np.int32) for f in ("M", "N", "K"))
            cap = int(sum(((n + 255) // 256) * ((k + 255) // 256) for n, k in zip(Ns, Ks))) + 2 * nwg + 1
            items = np.zeros((cap, 8), dtype=np.int32)
            first = np.zeros(nwg + 1, dtype=np.int32)
            n_items = _lib.load().climb_tn_grouped_plan(len(pending), Ms.ctypes.data, Ns.ctypes.data, Ks.ctypes.data, nwg, items.ctypes.data, cap, first.ctypes.data)
            if n_items <= 0:
                raise RuntimeError(f"climb_tn_grouped_plan failed: {_lib.error_string(n_items)} (code {n_items})")
            dev = self.device
            plan = dict(probs=torch.from_numpy(rec.view(np.uint8).copy()).to(dev), items=torch.from_numpy(items[:n_items].copy()).to(dev),
                        first=torch.from_numpy(first).to(dev), nwg=nwg, flops=float(sum(2.0 * M * N * K for _, _, _, _, M, N, K in pending)),
                        ragged=int(any((N % 256) or (K % 256) for _, _, _, _, M, N, K in pending)),
                        keep=[(dY, X) for dY, X, *_ in pending],
                        names=[w for _, _, w, *_ in pending], shapes=[(N, K) for _, _, _, _, M, N, K in pending],
                        # problems ALL of whose tiles are whole tiles (no stream-K share): the ones the optimizer may be fused into
                        whole=[bool(i not in set(int(x) for x in items[:n_items][items[:n_items, 5] == 1, 0])) for i in range(len(pending))], opts={}, split=sp)
            if len(ws.dw_plans) >= 16:           # requires_grad patterns / group sizes seen on this shape: bounded
                ws.dw_plans.pop(next(iter(ws.dw_plans)))
            ws.dw_plans[key] = plan
        if self.defer_dw and self.grad_ready_hook is None and self.loss_scale == 1.0 and not plan["ragged"]:
            self._dw_deferred.append((ws, plan))          # FusedAdamW.step() (or materialize_dw()) launches it
        else:
            self._launch_dw_plan(plan)
            self._grad_extra = True
        pending.clear()

    def _covered(self, name: str, numel: int):
        """layout tensors inside the flat range a weight-gradient problem writes (the fused QKV product covers query, key and value weights)"""
        lo = self.layout.offset[name]
        segs = getattr(self, "_segs", None)
        if segs is None:
            segs = self._segs = self.layout.segments()
        return [n for n, start, length in segs if lo <= start < lo + numel]

    def fused_dw_adamw(self, opt_m: torch.Tensor, opt_v: torch.Tensor, eligible, adam_row, ewc=None) -> set:
        """The held-back weight-gradient launches with AdamW in their epilogue (csrc/gemm_bf16_tnp.hip).  `eligible(name) -> bool`: tensors the
        optimizer updates THIS step with the constants `adam_row` (8 floats: lr, wd, beta1, beta2, eps, 1 - beta1^t, 1 - beta2^t, gradient scale).
        Returns the names that were updated here (the flat pass must skip them; their transposed shadows are fresh too)."""
        import numpy as np
        held, self._dw_deferred = self._dw_deferred, []
        done = set()
        row = np.ascontiguousarray(adam_row, dtype=np.float32)
        for ws, plan in held:
            cover = plan.setdefault("cover", [self._covered(n, N * K) for n, (N, K) in zip(plan["names"], plan["shapes"])])
            flags = tuple(bool(w and n in self._t_off and all(eligible(c) for c in cv)) for n, w, cv in zip(plan["names"], plan["whole"], cover))
            key = (opt_m.data_ptr(), opt_v.data_ptr(), flags)
            opts = plan["opts"].get(key)
            if opts is None:
                rec = np.zeros(len(flags), dtype=[("p", "<u8"), ("m", "<u8"), ("v", "<u8"), ("s", "<u8"), ("st", "<u8"), ("ldt", "<i8"), ("fused", "<i4"), ("pad", "<i4")])
                assert rec.dtype.itemsize == 56
                for r, n, (N, K), f in zip(rec, plan["names"], plan["shapes"], flags):
                    o = self.layout.offset[n]
                    r["p"], r["m"], r["v"] = self.p(n), opt_m.data_ptr() + 4 * o, opt_v.data_ptr() + 4 * o
                    r["s"], r["st"], r["ldt"], r["fused"] = self.sp(n), (self.spt(n) if n in self._t_off else 0), N, int(f)
                opts = torch.from_numpy(rec.view(np.uint8).copy()).to(self.device)
                if len(plan["opts"]) >= 8:
                    plan["opts"].pop(next(iter(plan["opts"])))
                plan["opts"][key] = opts
            if plan.get("split"):          # (r06) the same epilogue writing the hi AND lo planes of both shadows; no EWC fold in this mode (FusedAdamW.step())
                if ewc is not None:
                    raise RuntimeError("the split mode's optimizer epilogue carries no EWC term: FusedAdamW.step() folds it into the flat pass")
                self._timed_call("gemm_split_tn", plan["flops"], "climb_gemm_split_tn_grouped_adamw", plan["probs"], plan["items"], plan["first"], plan["nwg"], opts,
                                 row.ctypes.data, 1 if self._grad_extra else 0, self.layout.total, self._shadow_t.numel() // 2, _stream())
            elif ewc is not None:          # (never with a non-zero gradient buffer or a ragged plan: FusedAdamW.step() / _dw_flush)
                self._timed_call("gemm_bf16_tn", plan["flops"], "climb_gemm_bf16_tn_grouped_adamw_ewc", plan["probs"], plan["items"], plan["first"], plan["nwg"],
                                 plan["ragged"], opts, row.ctypes.data, 0, self.flat, ewc["star"], ewc["fisher"], ewc["lam"], ewc["loss"], _stream())
            else:
                self._timed_call("gemm_bf16_tn", plan["flops"], "climb_gemm_bf16_tn_grouped_adamw", plan["probs"], plan["items"], plan["first"], plan["nwg"], plan["ragged"],
                                 opts, row.ctypes.data, 1 if self._grad_extra else 0, _stream())
            for cv, f in zip(cover, flags):
                if f:
                    done.update(cv)
        if done:
            self._fused_consumed = True
        return done

    def _red_flush(self, ws: Workspace, pending: list):
        """the {dgamma, dbeta, bias} reductions recorded by a group's LayerNorm backwards, as one launch"""
        rg = self.requires_grad
        live = [p for p in pending if any(n is not None and rg[n] for n in p[3])]      # a frozen base (adapters, frozen layers): nothing to reduce
        pending.clear()
        pending.extend(live)
        if not pending:
            return
        import numpy as np
        key = tuple((part.data_ptr(), nblk, ncols, tuple(n if (n is not None and rg[n]) else None for n in names)) for part, nblk, ncols, names in pending)
        plan = ws.red_plans.get(key)
        if plan is None:
            rec = np.zeros(len(pending), dtype=[("part", "<u8"), ("stride", "<i8"), ("out", "<u8", (3,)), ("nblk", "<i4"), ("ncols", "<i4")])
            assert rec.dtype.itemsize == 48
            for r, (part, nblk, ncols, names) in zip(rec, pending):
                r["part"], r["stride"], r["nblk"], r["ncols"] = part.data_ptr(), 3 * ncols, nblk, ncols
                r["out"] = [self.g(n) if (n is not None and rg[n]) else 0 for n in names]
            plan = dict(segs=torch.from_numpy(rec.view(np.uint8).copy()).to(self.device), n=len(pending), cols=max(p[2] for p in pending))
            if len(ws.red_plans) >= 16:
                ws.red_plans.pop(next(iter(ws.red_plans)))
            ws.red_plans[key] = plan
        _lib.call("climb_colreduce_batched", plan["segs"], plan["n"], plan["cols"], _stream())
        pending.clear()

    # ------------------------------------------------------------------ encoder backward
    def _trainable_runs(self, lo, hi):
        """maximal runs of TRAINABLE tensors inside the flat range [lo, hi): what a data-parallel reducer has to carry (a frozen base
        under adapters, or frozen heads, contribute nothing) and what the optimizer may touch"""
        segs = getattr(self, "_segs", None)
        if segs is None:
            segs = self._segs = self.layout.segments()
        runs = []
        for name, start, length in segs:
            if start < lo or start >= hi or not self.requires_grad[name] or name in self._unused:
                continue
            if runs and runs[-1][1] == start:
                runs[-1][1] = start + length
            else:
                runs.append([start, start + length])
        return runs

    def _ready(self, lo, hi):
        for a, b in self._trainable_runs(lo, hi):
            hook = self.grad_ready_hook
            # fp16 operands: this backward's contributions (and, pre-scaled, earlier sums) carry the loss scale.  Without a data-parallel hook
            # nothing looks at the range before the backward ends: ONE pass over the buffer in finish_scaled_backward() divides it out (same
            # box, ms/step: no unscale 12.14, one pass at the end 12.30, a pass per finished range 12.37; bf16 11.91).  A reducer with a half payload
            # wants the range still scaled and divides the scale out itself; any other reducer gets it unscaled here.
            keep_scaled = hook is not None and getattr(getattr(hook, "__self__", None), "takes_scaled", False)
            if self.loss_scale != 1.0:
                if hook is None and _UNSCALE_MODE == "range":
                    _lib.call("climb_scale_f32", self.grad[a:b], b - a, 1.0 / self.loss_scale, _stream())
                elif hook is None:
                    self._unscale_pending = _UNSCALE_MODE != "none"
                elif not keep_scaled:
                    if self._prescaled:
                        raise NotImplementedError("fp16 operands: accumulating onto earlier gradient sums under data parallelism with an fp32 payload")
                    _lib.call("climb_scale_f32", self.grad[a:b], b - a, 1.0 / self.loss_scale, _stream())
            self.touched.append((a, b))
            if hook is not None:
                hook(a, b)

    def encoder_backward(self, dpooled: torch.Tensor, first_layer: int = 0, embeddings: bool = True, dpooled_is_dpre: bool = False):
        """Accumulates parameter gradients into the flat grad buffer.  `first_layer` / `embeddings` let frozen prefixes
        (REF/modeling/vilt.py:126-144) be skipped entirely."""
        cfg, lay = self.cfg, self.layout
        sv = self.saved
        ws: Workspace = sv["ws"]
        B, T, M = ws.B, ws.T, ws.M
        H, Fd = cfg["hidden"], cfg["ffn"]
        st = _stream()
        adt = self.adt
        lnb = _lib.query("climb_layernorm_bwd_rows_per_block")
        nlnb = (M + lnb - 1) // lnb
        rg = self.requires_grad
        ad = sv.get("adapter")
        r = self.layout.adapters[ad] if ad is not None else 0
        # pooler: pooled = tanh(clsn Wp^T + b)
        if dpooled_is_dpre:              # the head's last product already carried x (1 - pooled^2) in its epilogue
            dpre = dpooled
        else:
            dpre = ws.dpre
            _lib.call("climb_elementwise", 2, dpooled, ws.pooled, dpre, B * H, 1.0, st)
        pw, pb = ENC + "pooler.dense.weight", ENC + "pooler.dense.bias"
        if rg[pw]:
            self._rank_update(dpre, H, ws.clsn, H, pw, B, H, H)
        if SKINNY and B <= 64:           # d(clsn) = dpre Wp, and the bias gradient = the column sums of its A operand, in the same launch
            self._skinny(dpre, H, self.p(pw), 1, H, ws.dclsn, H, B, H, H, acol=self.g(pb) if rg[pb] else None)
        else:
            if rg[pb]:
                self._gemm_f32(dpre, 1, H, ws.ones, 0, 1, self.g(pb), 1, H, 1, B, beta=1.0)
            self._gemm_f32(dpre, H, 1, self.p(pw), 1, H, ws.dclsn, H, B, H, H)
        # final LayerNorm (row 0 of every sequence); all other rows of d(x_L) are zero
        ws.dres.zero_()
        prune = bool(sv.get("cls_only"))
        SH = ws.S_pad * H
        xL, ldxL = (ws.xLc, H) if prune else (ws.x[cfg["layers"]], SH)
        _lib.call("climb_layernorm_bwd", ws.dclsn, H, F32, xL, ldxL, ws.fmean, ws.frstd, self.p(ENC + "layernorm.weight"), None, 0,
                  ws.dres, SH, None, 0, ws.part, B, H, st)
        last = f"{ENC}encoder.layer.{cfg['layers'] - 1}."
        prune = prune and first_layer < cfg["layers"]
        # d(x_L) is non-zero on the B [CLS] rows only, so the column sums the kernel leaves next to dgamma / dbeta ARE the gradient of the last
        # layer's output bias (no pass over the M - B zero rows)
        last_bias = last + "output.dense.bias" if (first_layer < cfg["layers"] and ad is None) else None
        self.reduce3(ws.part, (B + lnb - 1) // lnb, H, ENC + "layernorm.weight", ENC + "layernorm.bias", last_bias)
        self._ready(*lay.top_range)
        # d(x_L): cast for the GEMMs (bf16 mode) + column sums for the last layer's output bias
        csr = _lib.query("climb_colsum_rows_per_block")
        self.materialize_dw()                    # (a backward on top of one whose weight gradients were held for an optimizer step that never came)
        G = self._dw_group_size(ws, ad)          # > 0: weight gradients are recorded per layer and launched per group of G layers
        pending, pending_red, group = [], [], []
        rgroup = self._ready_group()
        if G:
            ws.ensure_deferred(self)
        RB = bool(G) and _RED_BATCH
        lnpart = (lambda k: ws.part_l[k]) if RB else (lambda k: ws.part)          # partial sums of LayerNorm backward k (2 per layer)
        red3 = (lambda part, *names: pending_red.append((part, nlnb, H, names))) if RB else (lambda part, *names: self.reduce3(part, nlnb, H, *names))
        dxc = (lambda i: ws.dx_c[i]) if G else (lambda i: ws.dres_c)          # 16-bit d(x_i) / d(h1_i) / d(u_i) / d(qkv_i): per layer when deferred
        dhc = (lambda i: ws.dh_c[i]) if G else (lambda i: ws.dres_c)
        du_ = (lambda i: ws.du_l[i]) if G else (lambda i: ws.du)
        dqkv_ = (lambda i: ws.dqkv_l[i]) if G else (lambda i: ws.dqkv_s if self.split else ws.dqkv)
        odt = self.odt
        no_cast = self.precision == "fp32" and not self.split          # the plain fp32 mode's GEMMs read the fp32 residual-gradient stream itself
        dw = (lambda *a, **k: self._dw_defer(pending, *a, **k)) if G else self.dw_async
        nL = cfg["layers"]
        if prune:       # d(x_L) is non-zero on the [CLS] rows alone: their 16-bit copy is a compact [B, H] operand
            if self.precision != "fp32":
                ws.dyc.copy_(ws.dres.view(B, ws.S_pad, H)[:, 0])
        elif self.split:                    # d(x_L) as a GEMM operand
            self.split_of(ws.dres, dxc(nL), M, H)
        elif self.precision != "fp32":      # the GEMMs' 16-bit copy of d(x_L): zeros, and the B rows that are not
            dxc(nL).zero_()
            dxc(nL).view(B, ws.S_pad, H)[:, 0].copy_(ws.dres.view(B, ws.S_pad, H)[:, 0])
        for i in range(cfg["layers"] - 1, first_layer - 1, -1):
            l = f"{ENC}encoder.layer.{i}."
            if prune and i == nL - 1:
                self._last_layer_backward_cls(ws, l, i)     # MLP, LayerNorm and out-projection on the B [CLS] rows; leaves d(ctx) in ws.dctx
            else:
                # MLP: x_{i+1} = h1 + W2 gelu(u) + b2,  u = W1 hn + b1
                if ad is None:
                    dy = dxc(i + 1)
                    dw(dy, ws.a[i], l + "output.dense.weight", M, H, Fd)
                else:   # x_{i+1} = h1 + y + up(silu(down(y))): d(y) = d(x_{i+1}) + down^T(silu'(z) * up^T d(x_{i+1}))
                    dy = self.adapter_backward(ws, f"{l}output.adapters.{ad}.", ws.so[i], ws.zo[i], ws.yo[i], M, H, r,
                                               dxc(i + 1), ws.dz_l[2 * i + 1] if G else ws.dz, dw)
                    dw(dy, ws.a[i], l + "output.dense.weight", M, H, Fd, l + "output.dense.bias", ws)
                du = du_(i)
                if self.split and SPLIT_FUSED_ACT and _lib.query_arg("climb_gemm_split_nt_takes_act", M, Fd, H):
                    wn = l + "output.dense.weight"
                    self._timed_call("gemm_split_nt", 2.0 * M * Fd * H, "climb_gemm_split_nt_act", dy, H, M * H, self.spt(wn), H, self._shadow_t.numel() // 2, None, 0, du, Fd, M * Fd,
                                     M, Fd, H, None, EPI_DGELU_SP, ws.u[i], Fd, _stream())
                elif self.split:          # d(gelu output) in fp32, then x gelu'(u) (exact erf form, like the fp32 mode) into the operand planes
                    self.linear_dx(dy, l + "output.dense.weight", ws.du_f, M, H, Fd)
                    self.split_of(ws.du_f, du, M, Fd, mode=2, aux=ws.u[i])
                else:
                    self.linear_dx(dy, l + "output.dense.weight", du, M, H, Fd, self.epi_dgelu, ws.u[i])
                dw(du, ws.hn[i], l + "intermediate.dense.weight", M, Fd, H, l + "intermediate.dense.bias", ws)
                self.linear_dx(du, l + "intermediate.dense.weight", ws.dhn, M, Fd, H)
                self.join_side()          # LN backward overwrites d(residual) that dW2 is reading
                _lib.call("climb_layernorm_bwd", ws.dhn, H, odt, ws.h1[i], H, ws.mean2[i], ws.rstd2[i], self.p(l + "layernorm_after.weight"),
                          ws.dres, H, ws.dres, H, None if no_cast else dhc(i), H, lnpart(2 * i + 1), M, H, st)
                red3(lnpart(2 * i + 1), l + "layernorm_after.weight", l + "layernorm_after.bias", l + "attention.output.dense.bias" if ad is None else None)
                # attention: h1 = x + Wo ctx + bo
                if ad is None:
                    dy = dhc(i)
                    dw(dy, ws.ctx_s[i] if self.split else ws.ctx[i], l + "attention.output.dense.weight", M, H, H)
                else:
                    dy = self.adapter_backward(ws, f"{l}attention.output.adapters.{ad}.", ws.sa[i], ws.za[i], ws.ya[i], M, H, r,
                                               dhc(i), ws.dz_l[2 * i] if G else ws.dz, dw)
                    dw(dy, ws.ctx_s[i] if self.split else ws.ctx[i], l + "attention.output.dense.weight", M, H, H, l + "attention.output.dense.bias", ws)
                self.linear_dx(dy, l + "attention.output.dense.weight", ws.dctx, M, H, H)
            dqkv = dqkv_(i)
            if self.split:          # d(qkv) leaves the kernel as the operand planes of the two QKV gradient GEMMs (nothing else reads it)
                _lib.call("climb_attn_delta", ws.dctx, ws.ctx[i], F32, ws.delta, B, ws.S_pad, cfg["heads"], st)
                _lib.call("climb_attn_bwd_split", ws.qkv[i], ws.key_bias, ws.dctx, ws.lse[i], ws.delta, None, dqkv, M * 3 * H, B, ws.S_pad, cfg["heads"], cfg["head_dim"], st)
            else:
                self.attn_bwd(ws.qkv[i], ws.key_bias, ws.dctx, ws.ctx[i], ws.lse[i], ws.delta, dqkv, B, ws.S_pad)
            # q/k/v weights and biases are adjacent: one [2304,768] weight-gradient GEMM + one [2304] bias reduction
            dw(dqkv, ws.xn[i], l + "attention.attention.query.weight", M, 3 * H, H, l + "attention.attention.query.bias", ws)
            need_dx = i > first_layer or embeddings
            if need_dx:
                self.linear_dx(dqkv, l + "attention.attention.query.weight", ws.dxn, M, 3 * H, H)
                self.join_side()      # LN backward overwrites d(residual) (read by dWo); next layer overwrites du / dqkv
                _lib.call("climb_layernorm_bwd", ws.dxn, H, odt, ws.x[i], H, ws.mean1[i], ws.rstd1[i], self.p(l + "layernorm_before.weight"),
                          ws.dres, H, ws.dres, H, None if no_cast else dxc(i), H, lnpart(2 * i), M, H, st)
                red3(lnpart(2 * i), l + "layernorm_before.weight", l + "layernorm_before.bias",
                     f"{ENC}encoder.layer.{i - 1}.output.dense.bias" if (i > first_layer and ad is None) else None)
            self.join_side()
            if not G:          # immediate weight gradients: the layer is final now, reported in the same groups the deferred path uses
                group.append(i)
                if len(group) >= rgroup or i == first_layer:
                    self._ready(lay.layer_range[min(group)][0], lay.layer_range[max(group)][1])
                    group = []
                continue
            group.append(i)
            last_group = i == first_layer
            if len(group) >= G and not last_group:
                self._dw_flush(ws, pending)
                self._red_flush(ws, pending_red)
                # the group's layers are adjacent in the flat buffer: ONE range (under data parallelism: one 4-layer collective, not four)
                self._ready(lay.layer_range[min(group)][0], lay.layer_range[max(group)][1])
                group = []
        do_emb = embeddings and first_layer == 0
        if do_emb:
            self.embedding_backward(ws, sv, pending if G else None)          # the patch projection's dW rides in the last group
        if G:
            self._dw_flush(ws, pending)
            self._red_flush(ws, pending_red)
            if group:
                self._ready(lay.layer_range[min(group)][0], lay.layer_range[max(group)][1])
        if do_emb:
            self._ready(*lay.embed_range)
        self.saved = None          # the activations are consumed: a later no-grad forward may use this workspace again

    def _last_layer_backward_cls(self, ws: Workspace, l: str, i: int):
        """Backward of the last layer's MLP, `layernorm_after` and attention out-projection on the B [CLS] rows (`cls_only_last`): d(x_L) is
        zero everywhere else, so every product below is the dense one with its all-zero rows left out.  In: d(x_L) in the [CLS] rows of
        ws.dres (fp32, row stride S_pad * H) and ws.dyc (16-bit modes).  Out: d(h1) in the same rows of ws.dres, d(ctx) in ws.dctx (zero
        off the [CLS] rows), the six parameter gradients accumulated."""
        cfg = self.cfg
        B, H, Fd = ws.B, cfg["hidden"], cfg["ffn"]
        SH, SF = ws.S_pad * H, ws.S_pad * Fd
        st = _stream()
        f32 = self.precision == "fp32"
        lnb = _lib.query("climb_layernorm_bwd_rows_per_block")
        dy, lddy = (ws.dres, SH) if f32 else (ws.dyc, H)
        # x_L = h1 + W2 gelu(u) + b2,  u = W1 hn + b1
        self._lin_dw_ld(dy, lddy, ws.ac, Fd, l + "output.dense.weight", B, H, Fd)
        self._lin_dx_ld(dy, lddy, l + "output.dense.weight", ws.duc, Fd, B, H, Fd, self.epi_dgelu, ws.uc, Fd)
        self._lin_dw_ld(ws.duc, Fd, ws.hnc, H, l + "intermediate.dense.weight", B, Fd, H, l + "intermediate.dense.bias", ws)
        self._lin_dx_ld(ws.duc, Fd, l + "intermediate.dense.weight", ws.dhnc, H, B, Fd, H)
        # hn = LN(h1): d(h1) = d(x_L) + LN'(d(hn)), in place in the [CLS] rows of the residual-gradient stream
        _lib.call("climb_layernorm_bwd", ws.dhnc, H, self.adt, ws.h1c, H, ws.mean2c, ws.rstd2c, self.p(l + "layernorm_after.weight"),
                  ws.dres, SH, ws.dres, SH, None if f32 else ws.dhcc, H, ws.part, B, H, st)
        self.reduce3(ws.part, (B + lnb - 1) // lnb, H, l + "layernorm_after.weight", l + "layernorm_after.bias", l + "attention.output.dense.bias")
        # h1 = x + Wo ctx + bo
        dh, lddh = (ws.dres, SH) if f32 else (ws.dhcc, H)
        self._lin_dw_ld(dh, lddh, ws.ctx[i], SH, l + "attention.output.dense.weight", B, H, H)
        ws.dctx.zero_()
        self._lin_dx_ld(dh, lddh, l + "attention.output.dense.weight", ws.dctx, SH, B, H, H)

    def adapter_fwd(self, a_: str, y, resid, z_pre, s_act, out, M, H, r):
        """out = resid + y + up(silu(down(y))), saving z = down(y) and s = silu(z) for the backward.  16-bit mode: one launch
        (`climb_adapter_fwd_bf16`) where the shape allows, else -- and always in the fp32 mode -- the two skinny GEMMs."""
        if self.split:          # y, z, s are fp32 here: the bottleneck's two skinny products on the exact-fp32 GEMM (the fp32 mode's launches)
            self._force_f32 = True
            try:
                self.linear_fwd(y, a_ + "adapter_down.0.weight", a_ + "adapter_down.0.bias", s_act, M, r, H, EPI_SILU, None, z_pre)
                self.linear_fwd(s_act, a_ + "adapter_up.weight", a_ + "adapter_up.bias", out, M, H, r, EPI_RESID2, resid, None, y, out_f32=True)
            finally:
                self._force_f32 = False
            return
        if self.precision != "fp32" and H % 128 == 0 and r % 16 == 0 and r <= 64 and _FUSED_ADAPTER:
            _lib.call("climb_adapter_fwd_bf16", y, H, resid, H, self.sp(a_ + "adapter_down.0.weight"), self.p(a_ + "adapter_down.0.bias"),
                      self.sp(a_ + "adapter_up.weight"), self.p(a_ + "adapter_up.bias"), z_pre, s_act, r, out, H, M, H, r, _stream())
            return
        self.linear_fwd(y, a_ + "adapter_down.0.weight", a_ + "adapter_down.0.bias", s_act, M, r, H, EPI_SILU, None, z_pre)
        self.linear_fwd(s_act, a_ + "adapter_up.weight", a_ + "adapter_up.bias", out, M, H, r, EPI_RESID2, resid, None, y, out_f32=True)

    def adapter_backward(self, ws: Workspace, a_: str, s_act, z_pre, y_in, M, H, r, dout_c=None, dz=None, dw=None):
        """Backward of out = resid + y + up(silu(down(y))) given d(out) in ws.dres (fp32) / `dout_c` (operand dtype).
        Accumulates the adapter's parameter gradients (`dw`: now, or recorded for the group's launch -- then `dout_c` and `dz` are per-layer
        buffers that stay valid until it) and returns d(y) = d(out) + down^T(silu'(z) * up^T d(out))."""
        if self.split and not self._force_f32:
            # the fp32 mode's launches on the fp32 residual gradient (exact-fp32 GEMM, immediate weight gradients), then d(y) as the operand planes the
            # sub-layer's input- and weight-gradient GEMMs read
            self._force_f32 = True
            try:
                self.adapter_backward(ws, a_, s_act, z_pre, y_in, M, H, r, ws.dres, ws.dz, self.linear_dw)
            finally:
                self._force_f32 = False
            self.split_of(ws.dy, ws.dy_s, M, H)
            return ws.dy_s
        dout_c = ws.dres_c if dout_c is None else dout_c
        dz = ws.dz if dz is None else dz
        dw = self.dw_async if dw is None else dw
        dw(dout_c, s_act, a_ + "adapter_up.weight", M, H, r, a_ + "adapter_up.bias", ws)
        if self.precision != "fp32" and H % 128 == 0 and r % 16 == 0 and r <= 64 and _FUSED_ADAPTER:
            _lib.call("climb_adapter_bwd_bf16", dout_c, H, ws.dres, H, self.spt(a_ + "adapter_up.weight"), self.spt(a_ + "adapter_down.0.weight"),
                      z_pre, dz, r, ws.dy, H, M, H, r, _stream())
        else:
            self.linear_dx(dout_c, a_ + "adapter_up.weight", dz, M, H, r, EPI_DSILU, z_pre)
            self.linear_dx(dz, a_ + "adapter_down.0.weight", ws.dy, M, r, H, EPI_RESID, ws.dres)
        dw(dz, y_in, a_ + "adapter_down.0.weight", M, r, H, a_ + "adapter_down.0.bias", ws)
        return ws.dy

    def embedding_backward(self, ws: Workspace, sv, pending: Optional[list] = None):
        cfg = self.cfg
        B, T, H = ws.B, ws.T, cfg["hidden"]
        st = _stream()
        e = ENC + "embeddings."
        rg = self.requires_grad
        ntypes = self.layout.shapes[e + "token_type_embeddings.weight"][0]
        g0 = cfg["image"] // cfg["patch"]
        _lib.call("climb_image_embed_bwd", ws.dres, ws.img_type, ws.dims if sv.get("var") else None, ws.dproj, self.adt,
                  self.g(e + "position_embeddings") if rg[e + "position_embeddings"] else None,
                  self.g(e + "cls_token") if rg[e + "cls_token"] else None, ws.part, B, T, ws.NP, ws.gw, g0, ws.S_pad, H, ntypes,
                  1 if ws.compact else 0, st)
        self.bias_grad_from_part(ws.part.data_ptr(), ntypes * H, ws.NP + 1, e + "token_type_embeddings.weight", ntypes * H)
        Kp = cfg["channels"] * cfg["patch"] ** 2
        dproj, a_patch = ws.dproj, ws.a_patch
        if self.split:
            self.split_of(ws.dproj, ws.dproj_s, B * ws.NP, H)
            dproj, a_patch = ws.dproj_s, ws.a_patch_s
        if pending is not None:
            self._dw_defer(pending, dproj, a_patch, e + "patch_embeddings.projection.weight", B * ws.NP, H, Kp, e + "patch_embeddings.projection.bias", ws)
        else:
            self.linear_dw(dproj, a_patch, e + "patch_embeddings.projection.weight", B * ws.NP, H, Kp, e + "patch_embeddings.projection.bias", ws)
        te = e + "text_embeddings."
        ie = sv.get("inputs_embeds")
        _lib.call("climb_embed_text_bwd", sv["input_ids"] if ie is None else None, sv["token_type_ids"],
                  self.p(te + "word_embeddings.weight") if ie is None else ie,
                  self.p(te + "token_type_embeddings.weight"), self.p(te + "position_embeddings.weight"), self.p(te + "LayerNorm.weight"),
                  ws.tmean, ws.trstd, ws.dres, B, T, ws.S_pad, H,
                  self.g(te + "word_embeddings.weight") if (rg[te + "word_embeddings.weight"] and ie is None) else None,
                  self.g(te + "position_embeddings.weight") if rg[te + "position_embeddings.weight"] else None,
                  ws.dpre, ws.part, ws.part2, 0 if ie is None else int(ie.shape[1]), st)
        etb = _lib.query("climb_embed_text_bwd_rows_per_block")
        nb = (B * T + etb - 1) // etb
        # [dgamma | dbeta | dmodality row 0 (text rows)] in one launch; the modality table starts with row 0
        self.reduce3(ws.part, nb, H, te + "LayerNorm.weight", te + "LayerNorm.bias", e + "token_type_embeddings.weight")
        if rg[te + "token_type_embeddings.weight"]:
            _lib.call("climb_colreduce", ws.part2, 2 * H, T, self.g(te + "token_type_embeddings.weight"), 2 * H, 1.0, st)

    # ------------------------------------------------------------------ task heads (fp32; REF/modeling/vilt.py:179-203)
    def _head_buffers(self, task_key: str, Bh: int, reuse: bool):
        """Scratch of the classification head and the loss for `Bh` rows.  `reuse` (the fused step: forward, loss and backward in one call)
        takes them from a per-(task, rows) cache -- no allocation in the step; the autograd-facing path gets fresh
        tensors, because its forward's buffers must survive until an unrelated backward."""
        key = (task_key, Bh)
        hb = self._head_ws.get(key) if reuse else None
        if hb is None:
            tc = self.task_cfgs[task_key]
            H, dev, f32 = self.cfg["hidden"], self.device, torch.float32
            D, NL = 2 * H, tc["num_labels"]
            ldl = _round_up(NL, 4)          # 16-byte aligned rows so the head GEMMs take the vector load paths (3129 -> 3132)
            lnb = _lib.query("climb_layernorm_bwd_rows_per_block")
            hb = dict(z=torch.empty((Bh, D), dtype=f32, device=dev), zn=torch.empty((Bh, D), dtype=f32, device=dev), gz=torch.empty((Bh, D), dtype=f32, device=dev),
                      mean=torch.empty((Bh,), dtype=f32, device=dev), rstd=torch.empty((Bh,), dtype=f32, device=dev),
                      logits=torch.zeros((Bh, ldl), dtype=f32, device=dev), dlogits=torch.zeros((Bh, ldl), dtype=f32, device=dev),
                      dg=torch.empty((Bh, D), dtype=f32, device=dev), dzn=torch.empty((Bh, D), dtype=f32, device=dev), dz=torch.empty((Bh, D), dtype=f32, device=dev),
                      part=torch.empty((((Bh + lnb - 1) // lnb) * 3 * D,), dtype=f32, device=dev), dx=torch.empty((Bh, tc.get("num_images", 1) * H), dtype=f32, device=dev),
                      ones=torch.ones((max(Bh, 8),), dtype=f32, device=dev), loss=torch.empty((), dtype=f32, device=dev))
            if reuse:
                if len(self._head_ws) >= 8:
                    self._head_ws.pop(next(iter(self._head_ws)))
                self._head_ws[key] = hb
        return hb

    def head_forward(self, task_key: str, pooled_in: torch.Tensor, training: bool, keep_mask: Optional[torch.Tensor] = None, reuse: bool = False):
        tc = self.task_cfgs[task_key]
        H = self.cfg["hidden"]
        st = _stream()
        hs = HeadState()
        hs.task, hs.x = task_key, pooled_in
        h = f"task_layer.{task_key}."
        dev = self.device
        if tc["model_type"] == "classification":
            Bh, Kin = pooled_in.shape
            D, NL = 2 * H, tc["num_labels"]
            hb = hs.buf = self._head_buffers(task_key, Bh, reuse)
            hs.z, hs.zn, hs.gz, hs.mean, hs.rstd = hb["z"], hb["zn"], hb["gz"], hb["mean"], hb["rstd"]
            ldl = _round_up(NL, 4)
            hs.logits = hb["logits"][:, :NL]
            if SKINNY:                    # r04 (csrc/heads.hip): 3 launches instead of 7 (no zero-fill, no separate activation)
                self._skinny(pooled_in, Kin, self.p(h + "0.weight"), Kin, 1, hs.z, D, Bh, D, Kin, self.p(h + "0.bias"))
                _lib.call("climb_layernorm_gelu_fwd", hs.z, D, self.p(h + "1.weight"), self.p(h + "1.bias"), self.cfg["head_ln_eps"], hs.zn, hs.gz, D,
                          hs.mean, hs.rstd, Bh, D, st)
                self._skinny(hs.gz, D, self.p(h + "3.weight"), D, 1, hs.logits, ldl, Bh, NL, D, self.p(h + "3.bias"))
                return hs.logits, hs
            self._gemm_f32(pooled_in, Kin, 1, self.p(h + "0.weight"), Kin, 1, hs.z, D, Bh, D, Kin, self.p(h + "0.bias"))
            _lib.call("climb_layernorm_fwd", hs.z, D, self.p(h + "1.weight"), self.p(h + "1.bias"), self.cfg["head_ln_eps"], hs.zn, D, F32,
                      hs.mean, hs.rstd, Bh, D, st)
            _lib.call("climb_elementwise", 0, hs.zn, None, hs.gz, Bh * D, 1.0, st)
            self._gemm_f32(hs.gz, D, 1, self.p(h + "3.weight"), D, 1, hs.logits, ldl, Bh, NL, D, self.p(h + "3.bias"))
            return hs.logits, hs
        # multi-choice: Dropout(0.1) -> Linear(768, 1) -> squeeze        pooled_in [b, nc, H]
        b, nc, _ = pooled_in.shape
        flat_in = pooled_in.reshape(b * nc, H)
        hs.keep = None
        if training:
            if keep_mask is None:
                keep_mask = (torch.rand((b * nc, H), device=dev) >= 0.1).to(torch.float32)
            hs.keep = keep_mask.reshape(b * nc, H).contiguous()
            hs.xd = torch.empty_like(flat_in)
            _lib.call("climb_elementwise", 3, flat_in, hs.keep, hs.xd, b * nc * H, 1.0 / 0.9, st)
        else:
            hs.xd = flat_in
        hs.logits = torch.empty((b, nc), dtype=torch.float32, device=dev)
        self._gemm_f32(hs.xd, H, 1, self.p(h + "1.weight"), H, 1, hs.logits, 1, b * nc, 1, H, self.p(h + "1.bias"))
        return hs.logits, hs

    def head_backward(self, hs: HeadState, dlogits: torch.Tensor, dtanh_of: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`dtanh_of`: the pooler's tanh output the head consumed ([rows, Kin]-compatible memory).  When given and honoured (hs.dx_is_dpre is set),
        the returned gradient is already multiplied by 1 - pooled^2: encoder_backward(..., dpooled_is_dpre=True) skips that pass."""
        hs.dx_is_dpre = False
        task_key = hs.task
        tc = self.task_cfgs[task_key]
        H = self.cfg["hidden"]
        st = _stream()
        h = f"task_layer.{task_key}."
        rg = self.requires_grad
        dev = self.device
        if tc["model_type"] == "classification":
            Bh, Kin = hs.x.shape
            D, NL = 2 * H, tc["num_labels"]
            ldl = dlogits.stride(0)
            hb = hs.buf
            ones = hb["ones"]
            if rg[h + "3.weight"]:
                self._rank_update(dlogits, ldl, hs.gz, D, h + "3.weight", Bh, NL, D)
            if SKINNY and Bh <= 64 and ldl % 4 == 0:
                # r04 (csrc/heads.hip): 5 launches instead of 13 -- d(zn) = (dlogits W3) gelu'(zn) with d(b3) = colsum(dlogits) from the same launch; the
                # LayerNorm backward's third partial sum IS d(b0); d(x) = dz W0 (x (1 - pooled^2) when the pooler's output is what the head consumed)
                dzn, dz, part, dx = hb["dzn"], hb["dz"], hb["part"], hb["dx"]
                self._skinny(dlogits, ldl, self.p(h + "3.weight"), 1, D, dzn, D, Bh, D, NL, epi=3, aux=hs.zn, ldaux=D,
                             acol=self.g(h + "3.bias") if rg[h + "3.bias"] else None)
                lnb = _lib.query("climb_layernorm_bwd_rows_per_block")
                _lib.call("climb_layernorm_bwd", dzn, D, F32, hs.z, D, hs.mean, hs.rstd, self.p(h + "1.weight"), None, 0, dz, D, None, 0, part, Bh, D, st)
                self.reduce3(part, (Bh + lnb - 1) // lnb, D, h + "1.weight", h + "1.bias", h + "0.bias")
                if rg[h + "0.weight"]:
                    self._rank_update(dz, D, hs.x, Kin, h + "0.weight", Bh, D, Kin)
                fuse_tanh = dtanh_of is not None and dtanh_of.numel() == Bh * Kin and dtanh_of.is_contiguous()
                self._skinny(dz, D, self.p(h + "0.weight"), 1, Kin, dx, Kin, Bh, Kin, D, epi=2 if fuse_tanh else 0, aux=dtanh_of if fuse_tanh else None, ldaux=Kin)
                hs.dx_is_dpre = fuse_tanh
                self._ready(*self.layout.head_range[task_key])
                return dx
            if rg[h + "3.bias"]:
                self._gemm_f32(dlogits, 1, ldl, ones, 0, 1, self.g(h + "3.bias"), 1, NL, 1, Bh, beta=1.0)
            dg = hb["dg"]
            self._gemm_f32(dlogits, ldl, 1, self.p(h + "3.weight"), 1, D, dg, D, Bh, D, NL)
            dzn = hb["dzn"]
            _lib.call("climb_elementwise", 1, dg, hs.zn, dzn, Bh * D, 1.0, st)
            dz = hb["dz"]
            lnb = _lib.query("climb_layernorm_bwd_rows_per_block")
            nb = (Bh + lnb - 1) // lnb
            part = hb["part"]
            _lib.call("climb_layernorm_bwd", dzn, D, F32, hs.z, D, hs.mean, hs.rstd, self.p(h + "1.weight"), None, 0, dz, D, None, 0, part, Bh, D, st)
            self.bias_grad_from_part(part.data_ptr(), 3 * D, nb, h + "1.weight", D)
            self.bias_grad_from_part(part.data_ptr() + 4 * D, 3 * D, nb, h + "1.bias", D)
            if rg[h + "0.weight"]:
                self._gemm_f32(dz, 1, D, hs.x, 1, Kin, self.g(h + "0.weight"), Kin, D, Kin, Bh, beta=1.0)
            if rg[h + "0.bias"]:
                self._gemm_f32(dz, 1, D, ones, 0, 1, self.g(h + "0.bias"), 1, D, 1, Bh, beta=1.0)
            dx = hb["dx"]
            self._gemm_f32(dz, D, 1, self.p(h + "0.weight"), 1, Kin, dx, Kin, Bh, Kin, D)
            self._ready(*self.layout.head_range[task_key])
            return dx
        b, nc = dlogits.shape
        n = b * nc
        ones = torch.ones((max(n, 8),), dtype=torch.float32, device=dev)
        dl = dlogits.reshape(n, 1)
        if rg[h + "1.weight"]:
            self._gemm_f32(dl, 1, 1, hs.xd, 1, H, self.g(h + "1.weight"), H, 1, H, n, beta=1.0)
        if rg[h + "1.bias"]:
            self._gemm_f32(dl, 1, 1, ones, 0, 1, self.g(h + "1.bias"), 1, 1, 1, n, beta=1.0)
        dx = torch.empty((n, H), dtype=torch.float32, device=dev)
        self._gemm_f32(dl, 1, 1, self.p(h + "1.weight"), 1, H, dx, H, n, H, 1)
        if hs.keep is not None:
            _lib.call("climb_elementwise", 3, dx, hs.keep, dx, n * H, 1.0 / 0.9, st)
        self._ready(*self.layout.head_range[task_key])
        return dx.view(b, nc, H)

    # ------------------------------------------------------------------ losses
    def loss_and_grad(self, task_key: str, logits: torch.Tensor, target: torch.Tensor, gscale: float = 1.0, hs: Optional["HeadState"] = None):
        """VQA: BCEWithLogits(mean)*num_labels (REF train_vqa.py:155-157); others: CrossEntropyLoss (train_nlvr2.py:80).  With the head
        state of a fused step the loss scalar and dlogits live in its cached buffers (the padding columns of dlogits were zeroed at
        allocation and are never written)."""
        st = _stream()
        Bh, NL = logits.shape
        hb = getattr(hs, "buf", None) if hs is not None else None
        if hb is not None and "dlogits" in hb and hb["dlogits"].shape == (Bh, logits.stride(0)):
            loss, dlogits = hb["loss"], hb["dlogits"][:, :NL]
        else:
            loss = torch.empty((), dtype=torch.float32, device=self.device)
            dlogits = torch.zeros((Bh, logits.stride(0)), dtype=torch.float32, device=self.device)[:, :NL]      # same padded rows as logits
        if task_key == "vqa":
            if self._bce_ws is None:
                self._bce_ws = torch.empty((_lib.query("climb_bce_workspace_floats"),), dtype=torch.float32, device=self.device)
            _lib.call("climb_bce_logits", logits, logits.stride(0), target, target.stride(0), dlogits, dlogits.stride(0), loss, self._bce_ws, Bh, NL,
                      gscale, st)
        else:
            _lib.call("climb_cross_entropy", logits, logits.stride(0), target, dlogits, dlogits.stride(0), loss, Bh, NL, gscale, st)
        return loss, dlogits

    # ------------------------------------------------------------------ EWC / Fisher on the contiguous encoder range
    def ewc_penalty(self, star: torch.Tensor, fisher: torch.Tensor, lam: float, add_grad: bool, gscale: float = 1.0) -> torch.Tensor:
        n = self.layout.encoder_end
        if self._ewc_ws is None:
            self._ewc_ws = torch.empty((_lib.query("climb_ewc_workspace_floats"),), dtype=torch.float32, device=self.device)
        out = torch.empty((), dtype=torch.float32, device=self.device)
        _lib.call("climb_ewc_penalty", self.flat, star, fisher, self.grad if add_grad else None, n, lam, gscale, self._ewc_ws, out, _stream())
        if add_grad:
            self._grad_extra = True          # (a held-back weight-gradient launch adds this term to its tile sums before the update)
        return out

    def fisher_accumulate(self, fisher: torch.Tensor):
        self.materialize_dw()
        _lib.call("climb_fisher_accum", fisher, self.grad, self.layout.encoder_end, _stream())
