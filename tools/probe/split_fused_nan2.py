import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import vilt_oracle as vo
import test_gpu_parity as tp
os.environ["CLIMB_AMD_FUSED_ADAMW"] = "1"
model, _ = tp.make_model(["vqa"], 42, precision="bf16x3")
model.train()
opt = model.create_optimizer({"lr": 1e-4, "weight_decay": 1e-2, "adam_epsilon": 1e-8})
opt.zero_grad()
eng = model._host.engine()
name = "vilt_encoder.vilt.encoder.layer.0.attention.output.dense.weight"
for s in range(2):
    enc = vo.synthetic_encodings(2, seed=300 + s)
    images, texts = tp.enc_to_inputs(enc)
    model.fused_forward_backward("vqa", images, texts, vo.synthetic_vqa_targets(2, seed=300 + s), optimizer=opt)
    torch.cuda.synchronize()
    print("step", s, "deferred plans", len(eng._dw_deferred), "grad_extra", eng._grad_extra, "grad_clean", eng._grad_clean, "dirty", eng._grad_dirty)
    p_before = eng.view(eng.flat, name).clone()
    g_before = eng.view(eng.grad, name).clone()
    print("  grad range non-zero:", int((g_before != 0).sum()), "non-finite", int((~torch.isfinite(g_before)).sum()))
    opt.step(); torch.cuda.synchronize()
    p = eng.view(eng.flat, name)
    o = eng.layout.offset[name]
    m, v = opt._m[o:o + p.numel()].view_as(p), opt._v[o:o + p.numel()].view_as(p)
    bad = ~torch.isfinite(p)
    print("  after step: non-finite p", int(bad.sum()), "of", p.numel(), " m finite", bool(torch.isfinite(m).all()), " v min", float(v.min()), "v max", float(v.max()))
    if bad.any():
        idx = bad.nonzero()
        print("  bad rows", sorted(set(idx[:, 0].tolist()))[:12], "bad cols", sorted(set(idx[:, 1].tolist()))[:12])
        print("  p_before at bad finite?", bool(torch.isfinite(p_before[bad]).all()), " m at bad", m[bad][:4].tolist(), " v at bad", v[bad][:4].tolist())
    opt.zero_grad()
