import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import vilt_oracle as vo
import test_gpu_parity as tp
name = sys.argv[1] if len(sys.argv) > 1 else "vilt_encoder.vilt.encoder.layer.0.attention.output.dense.weight"
out = {}
for fused in ("1", "0"):
    os.environ["CLIMB_AMD_FUSED_ADAMW"] = fused
    model, _ = tp.make_model(["vqa"], 42, precision="bf16x3")
    model.train()
    opt = model.create_optimizer({"lr": 1e-4, "weight_decay": 1e-2, "adam_epsilon": 1e-8})
    opt.zero_grad()
    eng = model._host.engine()
    enc = vo.synthetic_encodings(2, seed=300)
    images, texts = tp.enc_to_inputs(enc)
    model.fused_forward_backward("vqa", images, texts, vo.synthetic_vqa_targets(2, seed=300), optimizer=opt)
    opt.step(); torch.cuda.synchronize()
    o = eng.layout.offset[name]; n = eng.layout.numel(name); shp = eng.layout.shapes[name]
    out[fused] = [t[o:o + n].view(shp).clone() for t in (eng.flat, opt._m, opt._v)] + [eng._shadow[o:o + n].view(shp).float().clone(), eng._shadow[eng.layout.total + o: eng.layout.total + o + n].view(shp).float().clone()]
for k, nm in enumerate(("p", "m", "v", "shadow hi", "shadow lo")):
    a, b = out["1"][k], out["0"][k]
    d = (a - b).abs()
    bad = d > 1e-6 + 1e-3 * b.abs()
    print(nm, "mismatches", int(bad.sum()), "of", a.numel(), "max diff", float(d.max()))
    if bad.any() and nm in ("p", "m", "v"):
        idx = bad.nonzero()[:6].tolist()
        print("   ", [(i, j, float(a[i, j]), float(b[i, j])) for i, j in idx])
        rows = sorted(set(bad.nonzero()[:, 0].tolist())); cols = sorted(set(bad.nonzero()[:, 1].tolist()))
        print("    rows", rows[:16], "... cols", cols[:16], "n rows", len(rows), "n cols", len(cols))
import numpy as np
a, b = out["1"][1], out["0"][1]
bad = ((a - b).abs() > 1e-6 + 1e-3 * b.abs())
bv = a[bad]
# does every wrong value occur somewhere in the reference m (a misplaced store) or in the reference of ANOTHER quantity?
refs = {"m": out["0"][1], "v": out["0"][2], "p": out["0"][0]}
for nm, r in refs.items():
    flat = r.flatten()
    hits = sum(int((flat - x).abs().min() < 1e-9) for x in bv[:40])
    print("wrong m values found in reference", nm, ":", hits, "of", min(40, bv.numel()))
# maybe they are (1 - beta1) * g of another element: g = m_ref / 0.1
g = out["0"][1] / 0.1
print("wrong m / 0.1 found among g:", sum(int(((g.flatten() - x / 0.1).abs().min()) < 1e-7) for x in bv[:40]))
i, j = bad.nonzero()[0].tolist()
print("first bad m at", (i, j), "value", float(a[i, j]), "ref", float(b[i, j]), " neighbours ref m row:", b[i, j - 2:j + 3].tolist())
pf = out["1"][0].flatten()
ii, jj = bad.nonzero(as_tuple=True)
for t in range(0, min(24, ii.numel())):
    x = a[ii[t], jj[t]]
    src = int((pf - x).abs().argmin())
    print("m[%d,%d] holds p[%d,%d]" % (int(ii[t]), int(jj[t]), src // a.shape[1], src % a.shape[1]))
av, bv2 = out["1"][2], out["0"][2]
badv = ((av - bv2).abs() > 1e-9 + 1e-3 * bv2.abs())
ii, jj = badv.nonzero(as_tuple=True)
for t in range(0, min(24, ii.numel())):
    x = av[ii[t], jj[t]]
    src = int((pf - x).abs().argmin())
    print("v[%d,%d]=%g holds p[%d,%d]=%g" % (int(ii[t]), int(jj[t]), float(x), src // a.shape[1], src % a.shape[1], float(pf[src])))
