"""debug: split attention backward vs the fp32 kernel on the engine's own tensors of the NLVR2 variable-resolution fixture"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import vilt_oracle as vo
from climb_amd import _lib
import test_gpu_parity as tp
z = np.load(os.path.join(ROOT, "tests/golden/nlvr2_b4_varres.npz"))
m = tp._meta(z)
b = int(m["b"])
sizes = [tuple(int(v) for v in r) for r in z["sizes"]]
model, P = tp.make_model(m["tasks"].split(","), int(m["wseed"]), precision="fp32")
e1 = vo.synthetic_varres_encodings(sizes, seed=int(m["dseed"]))
texts = dict(input_ids=e1["input_ids"][:b], token_type_ids=e1["token_type_ids"][:b], attention_mask=e1["attention_mask"][:b])
images = dict(pixel_values=e1["pixel_values"], pixel_mask=e1["pixel_mask"])
model.train()
model.fused_forward_backward("nlvr2", images, texts, torch.from_numpy(z["labels"]))
eng = model._host.engine()
ws = eng.last_ws
B, S_pad, M, H = ws.B, ws.S_pad, ws.M, 768
print("B", B, "S_pad", S_pad, "S", ws.S, "compact", ws.compact)
st = torch.cuda.current_stream().cuda_stream
qkv, kb, ctx, lse = ws.qkv[0], ws.key_bias, ws.ctx[0], ws.lse[0]
print("masked keys per row", (kb < -1e30).sum(1).tolist())
g = torch.Generator(device="cuda").manual_seed(0)
dctx = torch.randn(M, H, device="cuda", generator=g)
delta = torch.empty_like(lse)
_lib.call("climb_attn_delta", dctx, ctx, 0, delta, B, S_pad, 12, st)
d32 = torch.empty(M, 3 * H, device="cuda")
_lib.call("climb_attn_bwd_f32", qkv, kb, dctx, lse, delta, d32, B, S_pad, 12, 64, st)
ds = torch.full((M, 3 * H), float("nan"), device="cuda")
dsp = torch.empty((2, M, 3 * H), dtype=torch.bfloat16, device="cuda")
_lib.call("climb_attn_bwd_split", qkv, kb, dctx, lse, delta, ds, dsp, M * 3 * H, B, S_pad, 12, 64, st)
torch.cuda.synchronize()
print("nan", int(torch.isnan(ds).sum()), "inf", int(torch.isinf(ds).sum()))
err = (ds - d32).abs()
print("max err", float(err.max()), "ref max", float(d32.abs().max()))
bad = (err > 1e-3 * d32.abs().max()).nonzero()
print("bad count", len(bad), bad[:10].tolist())
if len(bad):
    rows = torch.unique(bad[:, 0])
    print("bad rows (b, s):", [(int(r) // S_pad, int(r) % S_pad) for r in rows[:40]])
    cols = torch.unique(bad[:, 1] // 768)
    print("bad thirds:", cols.tolist())
c2 = torch.empty(M, H, device="cuda"); l2 = torch.empty_like(lse)
_lib.call("climb_attn_fwd_split", qkv, kb, c2, None, 0, l2, B, S_pad, 12, 64, st)
print("fwd ctx err", float((c2 - ctx).abs().max()), "lse err", float((l2 - lse).abs().max()))
