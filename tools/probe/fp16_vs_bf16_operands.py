"""How much of the bf16 mode's error is the 8-bit mantissa?  HF ViltModel (random init, the benchmark's shapes) under torch autocast with
bf16 and with fp16 GEMM operands against fp32, on the GPU box.  (torch here is the measuring instrument, not the product.)"""
import torch
from transformers import ViltConfig, ViltModel

torch.manual_seed(0)
dev = torch.device("cuda:0")
m = ViltModel(ViltConfig()).to(dev).eval()
B, T = 64, 40
ids = torch.randint(0, 30522, (B, T), device=dev)
pix = torch.randn(B, 3, 384, 384, device=dev)
kw = dict(input_ids=ids, pixel_values=pix, attention_mask=torch.ones(B, T, dtype=torch.long, device=dev), pixel_mask=torch.ones(B, 384, 384, dtype=torch.long, device=dev))


def run(dtype):
    torch.manual_seed(1)          # visual_embed draws a patch permutation
    with torch.no_grad():
        if dtype is None:
            return m(**kw).pooler_output.float()
        with torch.autocast("cuda", dtype=dtype):
            return m(**kw).pooler_output.float()


ref = run(None)
for name, dt in [("bf16", torch.bfloat16), ("fp16", torch.float16)]:
    d = run(dt) - ref
    print(f"{name}: pooled max|d|/max|ref| {d.abs().max() / ref.abs().max():.2e}   rms(d)/rms(ref) {d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt():.2e}")
