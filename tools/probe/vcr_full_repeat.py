"""How much does the bf16 mode's vcr_b16 gradient-norm error move from run to run, alone and beside another process that keeps the GPU busy?"""
import os, sys, subprocess, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "16")
import numpy as np
import torch
from tests import test_gpu_parity as tp

def once(fname):
    z = np.load(os.path.join(ROOT, "tests", "golden", fname))
    e = tp.full_size_errors(z, tp.H16)
    return e["grad_norm_max"], e["pooled"], e["logits"]

if len(sys.argv) > 1 and sys.argv[1] == "load":
    a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
    t0 = time.time()
    while time.time() - t0 < float(sys.argv[2]):
        for _ in range(50):
            a @ a
        torch.cuda.synchronize()
    sys.exit(0)
for fname in ("vcr_b16.npz", "vqa_b64.npz", "nlvr2_b32.npz"):
    print(fname, "alone ", ["%.5f" % once(fname)[0] for _ in range(5)], flush=True)
p = subprocess.Popen([sys.executable, __file__, "load", "60"])
time.sleep(8)
for fname in ("vcr_b16.npz", "vqa_b64.npz", "nlvr2_b32.npz"):
    print(fname, "beside a GPU hog", ["%.5f" % once(fname)[0] for _ in range(5)], flush=True)
p.wait()
