"""Which gradients of the vcr_b16 step differ between two runs of the same process, per arithmetic mode (and do pooled / logits)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "16")
import numpy as np
import torch
from tests import test_gpu_parity as tp

def run(fname, precision, train):
    z = np.load(os.path.join(ROOT, "tests", "golden", fname))
    m = tp._meta(z)
    task, images, texts, target = tp._full_size_inputs(z)
    model, _ = tp.make_model(m["tasks"].split(","), int(m["wseed"]), precision=precision)
    model.train() if train else model.eval()
    loss, (pooled, logits), _, _ = model.fused_forward_backward(task, images, texts, target)
    G = tp.grads_of(model)
    return float(loss), pooled.detach().float().cpu(), logits.detach().float().cpu(), G

for fname, train in (("vcr_b16.npz", False), ("vqa_b64.npz", False), ("vqa_b64.npz", True)):
    for precision in ("fp32", tp.H16, "bf16x3"):
        a, b = run(fname, precision, train), run(fname, precision, train)
        diff = [(float((a[3][n] - b[3][n]).norm() / (b[3][n].norm() + 1e-30)), n) for n in a[3] if not torch.equal(a[3][n], b[3][n])]
        diff.sort(reverse=True)
        print(f"{fname} train={train} {precision}: loss equal {a[0] == b[0]}, pooled equal {torch.equal(a[1], b[1])}, logits equal {torch.equal(a[2], b[2])}, "
              f"{len(diff)} of {len(a[3])} gradients differ; worst {diff[:3]}", flush=True)
