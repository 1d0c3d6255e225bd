"""LayerNorm forward / backward at the step's shape (M = 12288 rows x 768): achieved HBM bandwidth.  Run on the GPU box: python tools/ln_bench.py"""
import sys; sys.path.insert(0,'/root/repo')
import torch
from climb_amd import _lib
from tools.gemm_bench import timeit, dev, st
M,C=12288,768
dy=torch.randn(M,C,device=dev).bfloat16(); x=torch.randn(M,C,device=dev); mean=torch.randn(M,device=dev); rstd=torch.rand(M,device=dev)+0.5
gamma=torch.randn(C,device=dev); dres=torch.randn(M,C,device=dev); dc=torch.empty(M,C,device=dev,dtype=torch.bfloat16)
part=torch.empty((M//16+1)*3*C,device=dev)
t=timeit(lambda: _lib.call("climb_layernorm_bwd", dy, C, 1, x, C, mean, rstd, gamma, dres, C, dres, C, dc, C, part, M, C, st()))
print(f"LN bwd {t*1e6:.1f} us  {(M*C*(2+4+4+4+2))/t/1e12:.2f} TB/s")
xo=torch.empty(M,C,device=dev,dtype=torch.bfloat16); mo=torch.empty(M,device=dev); ro=torch.empty(M,device=dev)
for rpw in (1, 2, 3, 1, 2, 3):          # rows per wave (climb_set_option 21), interleaved twice
    _lib.call("climb_set_option", 21, rpw)
    t=timeit(lambda: _lib.call("climb_layernorm_fwd", x, C, gamma, gamma, 1e-12, xo, C, 1, mo, ro, M, C, st()))
    print(f"LN fwd, {rpw} row(s) per wave: {t*1e6:.1f} us  {(M*C*(4+2))/t/1e12:.2f} TB/s")
