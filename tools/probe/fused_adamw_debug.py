"""debug aid: one optimizer step with / without the update in the weight-gradient epilogue, per-tensor differences"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.test_gpu_parity import make_model, _rand_batch
dev = torch.device("cuda:0")
res = {}
for fused in (False, True):
    torch.manual_seed(0); random.seed(0)
    model, P = make_model(["vqa", "nlvr2"], 42, precision="bf16")
    model.train()
    opt = model.create_optimizer({"lr": 1e-3, "weight_decay": 1e-2, "adam_epsilon": 1e-8})
    opt.zero_grad()
    eng = model._host.engine()
    pixels, texts, target = _rand_batch(64, 900, dev)
    model.fused_forward_backward("vqa", pixels, texts, target, None, optimizer=opt if fused else None)
    if fused:
        for ws, plan in eng._dw_deferred:
            print("deferred plan:", len(plan["names"]), "problems, whole:", sum(plan["whole"]), "ragged", plan["ragged"])
    opt.step()
    eng.refresh_shadow()
    torch.cuda.synchronize()
    res[fused] = {n: (eng.view(eng.flat, n).clone(), eng.view(opt._m, n).clone(), eng.view(opt._v, n).clone(), eng.view(eng._shadow, n).clone()) for n in eng.layout.offset}
    res[fused]["__st"] = {n: eng._shadow_t[eng._t_off[n]:eng._t_off[n] + eng.layout.numel(n)].clone() for n in eng._t_off}
    del model, opt
nbad = 0
for n in res[False]:
    if n == "__st":
        continue
    a, b = res[False][n], res[True][n]
    d = [int((x != y).sum()) for x, y in zip(a, b)]
    if any(d):
        nbad += 1
        if nbad <= 12:
            x, y = a[0].double(), b[0].double()
            bad = (a[0] != b[0]).nonzero()
            print(n, tuple(a[0].shape), "differ p/m/v/s:", d, "max |dp|", float((x - y).abs().max()), "first bad idx", bad[:3].tolist(), "last", bad[-2:].tolist())
for n in res[False]["__st"]:
    d = int((res[False]["__st"][n] != res[True]["__st"][n]).sum())
    if d and nbad <= 14:
        print("transposed shadow", n, d)
print("tensors that differ:", nbad)
