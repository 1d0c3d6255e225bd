"""Per-workgroup timeline of the forward attention kernel (AB_TRACE build of attention_bf16.hip; see tools/attn_probe.sh)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from climb_amd import _lib

dev = torch.device("cuda:0")
B, heads, d, S_pad = 64, 12, 64, int(os.environ.get("S_PAD", 192))
H = heads * d
st = lambda: torch.cuda.current_stream().cuda_stream
qkv = torch.randn(B * S_pad, 3 * H, device=dev).bfloat16()
bias = torch.zeros(B, S_pad, device=dev)
ctx = torch.empty(B * S_pad, H, device=dev, dtype=torch.bfloat16)
lse = torch.empty(B, heads, S_pad, device=dev)
buf = torch.zeros(B * heads, 16, device=dev, dtype=torch.int64)
lib = ctypes.CDLL(os.environ["CLIMB_AMD_LIB"])
lib.climb_attn_set_trace.argtypes = [ctypes.c_void_p]
assert lib.climb_attn_set_trace(buf.data_ptr()) == 0
for _ in range(3):
    _lib.call("climb_attn_fwd_bf16", qkv, bias, ctx, lse, B, S_pad, heads, d, st())
torch.cuda.synchronize()
t = buf.cpu().double() * 0.01          # 100 MHz -> us
t0 = t[:, 0].min()
t = t - t0
names = ["start", "Q rows"] + [f"blk{k} ready" for k in range(S_pad // 32)]
cols = [0, 1] + [2 + k for k in range(min(10, S_pad // 32))] + [12, 13, 14]
names = names[:len(cols) - 3] + ["loop end", "stores issued", "stores done"]
print(f"{'event':>14} {'min':>7} {'p10':>7} {'median':>7} {'p90':>7} {'max':>7}   (us after the first workgroup's start; wave 0 of {B*heads} workgroups)")
for n, c in zip(names, cols):
    v = t[:, c].sort().values
    q = lambda f: v[int(f * (len(v) - 1))].item()
    print(f"{n:>14} {q(0):7.2f} {q(.1):7.2f} {q(.5):7.2f} {q(.9):7.2f} {q(1):7.2f}")
