"""Row F1 measurement: the image half of the input pipeline for one BASELINE batch (64 COCO-sized uint8 images), host
ViltProcessor (what the reference runs on its training thread, 1 thread) next to the device pipeline (raw-byte H2D + two HIP
stages).  Run on the GPU box:  python tools/input_bench.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image

from climb_amd.data import DeviceImagePipeline

rng = np.random.default_rng(0)
shapes = [(480, 640) if i % 2 else (int(rng.integers(330, 641)), int(rng.integers(330, 641))) for i in range(64)]
imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in shapes]
raw = sum(a.size for a in imgs)
dev = torch.device("cuda:0")
pipe = DeviceImagePipeline(dev)
out = pipe(imgs)
torch.cuda.synchronize()
out_bytes = out["pixel_values"].numel() * 4 + out["pixel_mask"].numel() * 8
t0 = time.perf_counter()
for _ in range(10):
    out = pipe(imgs)
torch.cuda.synchronize()
t_dev = (time.perf_counter() - t0) / 10
# device stages alone (inputs resident): replay the two launches
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(10):
    pipe(imgs)
e1.record()
torch.cuda.synchronize()
print(f"device pipeline: {t_dev*1e3:7.2f} ms per 64-image batch = {64/t_dev:8.0f} images/s   (raw in {raw/1e6:.1f} MB, tensors out {out_bytes/1e6:.1f} MB; "
      f"GPU-side span {e0.elapsed_time(e1)/10:.2f} ms incl. host planning)")
try:
    from transformers.models.vilt.image_processing_pil_vilt import ViltImageProcessorPil
    proc = ViltImageProcessorPil()
    pil = [Image.fromarray(a) for a in imgs]
    torch.set_num_threads(1)
    t0 = time.perf_counter()
    enc = proc(pil, return_tensors="pt")
    pv, pm = enc["pixel_values"].to(dev), enc["pixel_mask"].to(dev)
    torch.cuda.synchronize()
    t_host = time.perf_counter() - t0
    same = torch.equal(pv, out["pixel_values"]) and torch.equal(pm, out["pixel_mask"])
    print(f"host ViltProcessor + H2D (1 thread): {t_host*1e3:7.2f} ms per batch = {64/t_host:8.0f} images/s   identical tensors: {same}")
except Exception as e:      # noqa: BLE001
    print("host processor unavailable:", type(e).__name__, e)
