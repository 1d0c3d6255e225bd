"""The frozen BERT text encoder of ViLT-BERT: `BertModel(...).last_hidden_state` as
REF/modeling/viltbert.py:115-121 (`get_bert_outputs`) computes it -- forward only, no gradient, 2 560 text rows at 64 sequences of 40
tokens (about a fifth of one ViLT forward) -- on the same HIP kernels as the ViLT encoder, in post-LayerNorm order
(HFB = transformers/models/bert/modeling_bert.py: embeddings :53-116, self-attention :139-220, add & norm :282-294, :340-352).

`BertParams` holds exactly `BertModel`'s parameters (names, shapes, registration order: its state_dict interchanges with transformers'
`bert-base-uncased`).  Because the weights are frozen they are PACKED once per weight version into GEMM-ready device buffers: the
q / k / v matrices of a layer become one [2304, 768] operand (one fused QKV GEMM, as in the ViLT encoder), with bf16 copies in the
throughput mode.

Train mode (r03).  The reference runs BERT under `no_grad` but never calls `bert.eval()` (REF/modeling/viltbert.py:115-120): while the
learner is in train mode BERT's 37 dropouts (p = 0.1: after the embedding LayerNorm, on the attention probabilities, on both sub-layer
outputs before their residual adds; HFB:112-116, eager_attention_forward, :289-293, :348-351) perturb the "frozen" features -- measured
with the reference: 57 % relative rms on the features, 60 % on one step's gradient (tests/golden/viltbert_train_dropout.json).  This module
follows `self.training` the same way: in train mode every dropout is applied, from masks drawn with the device's generator (the same
distribution; torch's CPU stream cannot be reproduced bit for bit on another device) or from masks the caller passes
(`dropout_masks`, how tests/golden/viltbert_vqa_b3_train.npz pins the arithmetic against the reference's own masks).  Eval mode computes
the deterministic features (tests/golden/viltbert_vqa_b3.npz)."""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib

F32, BF16 = 0, 1
EPI_NONE, EPI_GELU, EPI_RESID = 0, 1, 2
BERT_CFG = dict(hidden=768, heads=12, head_dim=64, ffn=3072, layers=12, vocab=30522, max_pos=512, type_vocab=2, ln_eps=1e-12, dropout=0.1)


def bert_param_shapes(cfg: dict = BERT_CFG) -> "OrderedDict[str, tuple]":
    H, Fd = cfg["hidden"], cfg["ffn"]
    s: "OrderedDict[str, tuple]" = OrderedDict()
    s["embeddings.word_embeddings.weight"] = (cfg["vocab"], H)
    s["embeddings.position_embeddings.weight"] = (cfg["max_pos"], H)
    s["embeddings.token_type_embeddings.weight"] = (cfg["type_vocab"], H)
    s["embeddings.LayerNorm.weight"] = (H,)
    s["embeddings.LayerNorm.bias"] = (H,)
    for i in range(cfg["layers"]):
        l = f"encoder.layer.{i}."
        for n in ("query", "key", "value"):
            s[l + f"attention.self.{n}.weight"] = (H, H)
            s[l + f"attention.self.{n}.bias"] = (H,)
        for n, shp in (("attention.output.dense.weight", (H, H)), ("attention.output.dense.bias", (H,)), ("attention.output.LayerNorm.weight", (H,)),
                       ("attention.output.LayerNorm.bias", (H,)), ("intermediate.dense.weight", (Fd, H)), ("intermediate.dense.bias", (Fd,)),
                       ("output.dense.weight", (H, Fd)), ("output.dense.bias", (H,)), ("output.LayerNorm.weight", (H,)), ("output.LayerNorm.bias", (H,))):
            s[l + n] = shp
    s["pooler.dense.weight"] = (H, H)
    s["pooler.dense.bias"] = (H,)
    return s


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _round_up(x, m):
    return (x + m - 1) // m * m


class BertParams(nn.Module):
    """Stand-in for `transformers.BertModel` holding its parameters (frozen) and running its forward on the HIP kernels."""

    def __init__(self, cfg: dict = BERT_CFG, precision: str = "bf16"):
        super().__init__()
        from .modeling.vilt import _build_tree, _Node
        self.cfg, self.precision = cfg, precision
        tree = _Node()
        _build_tree(tree, bert_param_shapes(cfg))
        for name, child in tree._modules.items():       # embeddings / encoder / pooler directly under this module: keys match BertModel's
            self.add_module(name, child)
        for p in self.parameters():
            p.requires_grad = False
        self._packed: Optional[Dict[str, torch.Tensor]] = None
        self._packed_version = None
        self._ws: Dict[tuple, Dict[str, torch.Tensor]] = {}

    def __deepcopy__(self, memo):
        new = BertParams(self.cfg, self.precision)
        new.load_state_dict(self.state_dict())
        return new.to(next(self.parameters()).device)

    def init_like_hf(self, seed: Optional[int] = None):
        gen = torch.Generator().manual_seed(seed) if seed is not None else None
        with torch.no_grad():
            for n, p in self.named_parameters():
                if n.endswith(".bias"):
                    p.zero_()
                elif n.endswith("LayerNorm.weight"):
                    p.fill_(1.0)
                else:
                    p.copy_(torch.empty(p.shape).normal_(0.0, 0.02, generator=gen))

    # ------------------------------------------------------------------ packed operands
    def _pack(self, dev):
        ver = sum(p._version for p in self.parameters())
        if self._packed is not None and self._packed_version == ver and self._packed["word"].device == dev:
            return self._packed
        sd = {n: p.detach().to(dev, torch.float32) for n, p in self.named_parameters()}
        L = self.cfg["layers"]
        pk: Dict[str, torch.Tensor] = {}
        e = "embeddings."
        pk["word"], pk["pos"], pk["type"] = sd[e + "word_embeddings.weight"].contiguous(), sd[e + "position_embeddings.weight"].contiguous(), sd[e + "token_type_embeddings.weight"].contiguous()
        pk["eln_w"], pk["eln_b"] = sd[e + "LayerNorm.weight"].contiguous(), sd[e + "LayerNorm.bias"].contiguous()
        pk["zero_h"] = torch.zeros(self.cfg["hidden"], dtype=torch.float32, device=dev)

        def stack(fn):
            return torch.stack([fn(f"encoder.layer.{i}.") for i in range(L)]).contiguous()
        pk["wqkv"] = stack(lambda l: torch.cat([sd[l + f"attention.self.{n}.weight"] for n in ("query", "key", "value")], 0))
        pk["bqkv"] = stack(lambda l: torch.cat([sd[l + f"attention.self.{n}.bias"] for n in ("query", "key", "value")], 0))
        for key, name in (("wo", "attention.output.dense.weight"), ("bo", "attention.output.dense.bias"), ("ln1_w", "attention.output.LayerNorm.weight"),
                          ("ln1_b", "attention.output.LayerNorm.bias"), ("w1", "intermediate.dense.weight"), ("b1", "intermediate.dense.bias"),
                          ("w2", "output.dense.weight"), ("b2", "output.dense.bias"), ("ln2_w", "output.LayerNorm.weight"), ("ln2_b", "output.LayerNorm.bias")):
            pk[key] = stack(lambda l, name=name: sd[l + name])
        if self.precision != "fp32":
            for key in ("wqkv", "wo", "w1", "w2"):
                src = pk[key]
                dst = torch.empty(src.shape, dtype=_lib.torch_h16(), device=dev)
                _lib.call("climb_cast_bf16", src, dst, src.numel(), _stream())
                pk[key + "_op"] = dst
        else:
            for key in ("wqkv", "wo", "w1", "w2"):
                pk[key + "_op"] = pk[key]
        self._packed, self._packed_version = pk, ver
        return pk

    def _workspace(self, dev, B, T):
        key = (str(dev), B, T)
        ws = self._ws.get(key)
        if ws is None:
            if len(self._ws) >= 3:
                self._ws.pop(next(iter(self._ws)))
            H, Fd, nh = self.cfg["hidden"], self.cfg["ffn"], self.cfg["heads"]
            Tp = _round_up(T, 32)
            M = B * Tp
            adt = torch.float32 if self.precision == "fp32" else _lib.torch_h16()

            def buf(shape, dt=torch.float32):
                return torch.zeros(shape, dtype=dt, device=dev)
            ws = dict(Tp=Tp, M=M, x=buf((M, H)), y=buf((M, H)), h=buf((M, H)), qkv=buf((M, 3 * H), adt), ctx=buf((M, H), adt), pre=buf((M, Fd), adt),
                      a=buf((M, Fd), adt), lse=buf((B, nh, Tp)), key_bias=buf((B, Tp)), mean=buf((M,)), rstd=buf((M,)), tmean=buf((B * T,)), trstd=buf((B * T,)))
            ws["xb"] = ws["x"] if self.precision == "fp32" else buf((M, H), adt)
            ws["hb"] = ws["h"] if self.precision == "fp32" else buf((M, H), adt)
            self._ws[key] = ws
        return ws

    # ------------------------------------------------------------------ kernels
    def _gemm(self, X, W, bias, Y, M, N, K, epi=EPI_NONE, aux=None, aux_out=None, out_f32=False):
        st = _stream()
        if self.precision == "fp32":
            _lib.call("climb_gemm_f32", X, K, 1, W, K, 1, Y, N, M, N, K, bias, epi, aux, N, aux_out, N, 0.0, None, 0, 0, st)
        else:
            _lib.call("climb_gemm_bf16_nt", X, K, W, K, Y, N, F32 if out_f32 else BF16, M, N, K, bias, epi, aux, N, aux_out, N, None, 0, st)

    def _ln(self, x, w, b, out, out_dt, ws, M):
        H = self.cfg["hidden"]
        _lib.call("climb_layernorm_fwd", x, H, w, b, self.cfg["ln_eps"], out, H, out_dt, ws["mean"], ws["rstd"], M, H, _stream())

    def draw_dropout_masks(self, B: int, T: int, dev) -> Dict[str, object]:
        """Keep-masks of one train-mode forward: {"emb": [B,T,H], "probs": L x [B,heads,T,T], "attn_out": L x [B,T,H], "ffn_out": L x [B,T,H]}
        (bool, True = kept), Bernoulli(1 - p) from the device's default generator."""
        cfg = self.cfg
        H, nh, L, keep = cfg["hidden"], cfg["heads"], cfg["layers"], 1.0 - cfg.get("dropout", 0.1)

        # Under data parallelism every rank seeds the device generator alike (the loaders, the replay memory and EWC's task draw depend on it), so
        # the ranks would draw IDENTICAL masks for their different shards -- dropout noise correlated across the global batch, unlike the single
        # process's.  The masks therefore come from a generator of their own, offset by the rank; one rank: the device's default stream.
        gen = None
        from . import parallel
        rank, world = parallel.rank_world()
        if world > 1:
            gen = getattr(self, "_dropout_gen", None)
            seed0 = torch.initial_seed()
            if gen is None or getattr(self, "_dropout_seed0", None) != seed0:
                # (re-)derived whenever the process seed changes: torch.manual_seed() between tasks or on resume moves the masks under data
                # parallelism exactly as it moves the single process's default stream
                gen = self._dropout_gen = torch.Generator(device=dev)
                gen.manual_seed((seed0 + 7919 * (rank + 1)) % (2 ** 63))
                self._dropout_seed0 = seed0

        def draw(*shape):
            return torch.rand(shape, device=dev, generator=gen) < keep
        return {"emb": draw(B, T, H), "probs": [draw(B, nh, T, T) for _ in range(L)], "attn_out": [draw(B, T, H) for _ in range(L)],
                "ffn_out": [draw(B, T, H) for _ in range(L)]}

    @torch.no_grad()
    def forward(self, input_ids, attention_mask, token_type_ids, dropout_masks: Optional[Dict[str, object]] = None) -> torch.Tensor:
        """[B, T] int64 x 3 -> last_hidden_state as a [B, roundup(T, 32), 768] fp32 buffer whose first T rows per sequence are valid
        (the ViLT engine's `inputs_embeds` operand; rows beyond T are scratch).  Train mode (`self.training`, or masks given): dropouts live."""
        dev = input_ids.device
        if dev.type != "cuda":
            raise RuntimeError("climb_amd.bert needs a HIP device; there is no CPU path in the product (oracle/bert_oracle.py is the CPU checker)")
        cfg = self.cfg
        B, T = input_ids.shape
        H, Fd, nh = cfg["hidden"], cfg["ffn"], cfg["heads"]
        pk = self._pack(dev)
        ws = self._workspace(dev, B, T)
        Tp, M = ws["Tp"], ws["M"]
        st = _stream()
        adt = F32 if self.precision == "fp32" else BF16
        p_drop = float(cfg.get("dropout", 0.1))
        if dropout_masks is None and self.training and p_drop > 0.0:
            dropout_masks = self.draw_dropout_masks(B, T, dev)
        drop = dropout_masks is not None
        if drop and Tp > 64:
            raise NotImplementedError(f"BERT train-mode dropout: {T} text tokens (the probability-dropout kernel holds <= 64 keys; CLiMB's max is 40)")
        dscale = 1.0 / (1.0 - p_drop)

        def row_mask(m):          # [B, T, H] bool -> fp32 [B, Tp, H] in the activation layout (padding rows: 1, they are scratch)
            out = torch.ones((B, Tp, H), dtype=torch.float32, device=dev)
            out[:, :T] = m.to(dev, torch.float32)
            return out

        def dropout_rows(buf, m):
            _lib.call("climb_elementwise", 3, buf, row_mask(m), buf, M * H, dscale, st)
        _lib.call("climb_key_bias", attention_mask, ws["key_bias"], B, T, T, Tp, st)
        x = ws["x"]
        x.zero_()                                       # padding rows: finite (zero) inputs, masked as keys
        _lib.call("climb_embed_text_fwd", input_ids, token_type_ids, pk["word"], pk["type"], pk["pos"], pk["eln_w"], pk["eln_b"], pk["zero_h"],
                  cfg["ln_eps"], x, B, T, Tp, H, ws["tmean"], ws["trstd"], 0, st)
        if drop:
            dropout_rows(x, dropout_masks["emb"])                                                                  # HFB:112-116
        attn = "climb_attn_fwd_f32" if self.precision == "fp32" else "climb_attn_fwd_bf16"
        for i in range(cfg["layers"]):
            if self.precision != "fp32":
                _lib.call("climb_cast_bf16", x, ws["xb"], M * H, st)
            self._gemm(ws["xb"], pk["wqkv_op"][i], pk["bqkv"][i], ws["qkv"], M, 3 * H, H)
            if drop:
                keep = dropout_masks["probs"][i].to(dev, torch.uint8).contiguous()
                _lib.call("climb_attn_fwd_dropout", ws["qkv"], ws["key_bias"], keep, ws["ctx"], adt, B, Tp, nh, cfg["head_dim"], T, dscale, st)
                self._gemm(ws["ctx"], pk["wo_op"][i], pk["bo"][i], ws["y"], M, H, H, out_f32=True)                 # dense, THEN dropout, then + x
                dropout_rows(ws["y"], dropout_masks["attn_out"][i])
                _lib.call("climb_elementwise", 5, ws["y"], x, ws["y"], M * H, 1.0, st)
            else:
                _lib.call(attn, ws["qkv"], ws["key_bias"], ws["ctx"], ws["lse"], B, Tp, nh, cfg["head_dim"], st)
                self._gemm(ws["ctx"], pk["wo_op"][i], pk["bo"][i], ws["y"], M, H, H, EPI_RESID, aux=x, out_f32=True)          # + x (HFB:290)
            self._ln(ws["y"], pk["ln1_w"][i], pk["ln1_b"][i], ws["h"], F32, ws, M)
            if self.precision != "fp32":
                self._ln(ws["y"], pk["ln1_w"][i], pk["ln1_b"][i], ws["hb"], BF16, ws, M)
            self._gemm(ws["hb"], pk["w1_op"][i], pk["b1"][i], ws["a"], M, Fd, H, EPI_GELU, aux_out=ws["pre"])
            if drop:
                self._gemm(ws["a"], pk["w2_op"][i], pk["b2"][i], ws["y"], M, H, Fd, out_f32=True)
                dropout_rows(ws["y"], dropout_masks["ffn_out"][i])
                _lib.call("climb_elementwise", 5, ws["y"], ws["h"], ws["y"], M * H, 1.0, st)
            else:
                self._gemm(ws["a"], pk["w2_op"][i], pk["b2"][i], ws["y"], M, H, Fd, EPI_RESID, aux=ws["h"], out_f32=True)          # + h (HFB:349)
            self._ln(ws["y"], pk["ln2_w"][i], pk["ln2_b"][i], x, F32, ws, M)
        return x.view(B, Tp, H)
