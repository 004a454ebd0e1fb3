#!/usr/bin/env python3
"""Timing of the fused backward of the thin expand units of the MobileNetV3-YOLO 512x512 bs-64 plan on bf16 storage (mny_pw_bnbwd_bf16: 16 -> 64 @256x256,
24 -> 72 @128x128): the wave form of csrc/gate.hip (pwe_sums_kernel + finalize + pwe_dgrad_kernel); MNY_NO_PWE=1: the fp32-MFMA kernels of pwgemm.hip.
    python tools/bench_pwe.py      -> ms per whole unit (all its launches)"""
import sys, os
sys.path.insert(0, '/root/repo')
import torch
from mobilenet_yolo_pytorch_amd import ops
def rnd(*s): return torch.randn(*s, device='cuda')
bf=torch.bfloat16
for M,K,N in ((4194304,16,64),(1048576,24,72)):
    x=rnd(1,1,M,K).to(bf); w=(rnd(N,K)*K**-0.5)
    xs,xh=1+0.2*rnd(K),0.3*rnd(K); gamma,beta=1+0.3*rnd(N),0.2*rnd(N)
    y,st=ops.pw_fwd((x,xs,xh,3),w.to(bf)); scale,shift,mean,invstd=ops.bn_finalize(st,M,gamma,beta)
    g=rnd(1,1,M,N).to(bf)
    f=lambda: ops.pw_bnbwd(g,y,scale,shift,3,mean,invstd,gamma,(x,xs,xh,3),w)
    for _ in range(3): f()
    torch.cuda.synchronize(); a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): f()
    b.record(); torch.cuda.synchronize(); print(M,K,N,'%.3f ms'%(a.elapsed_time(b)/10), flush=True)
