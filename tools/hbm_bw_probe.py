#!/usr/bin/env python3
"""What torch's own streaming kernels reach on this GPU for 1-4 HBM streams of 3 GB (the practical ceiling the LayerNorm passes are
measured against: copy 4.8, add 6.1, addcmul 5.9, sum 4.0 TB/s on MI355X; ln_bwd moves its 4 streams at 5.1-5.5, ln_stats its one at 5.3).
    python tools/hbm_bw_probe.py"""
import torch
dev=torch.device('cuda:0')
n=120*96*256*256
a=torch.randn(n,device=dev); b=torch.randn(n,device=dev); c=torch.randn(n,device=dev); d=torch.empty(n,device=dev)
def t(fn,reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps
ms=t(lambda: d.copy_(a)); print('copy  (2 streams)', ms, 2*n*4/ms/1e9,'TB/s')
ms=t(lambda: torch.add(a,b,out=d)); print('add   (3 streams)', ms, 3*n*4/ms/1e9,'TB/s')
ms=t(lambda: torch.addcmul(a,b,c,out=d)); print('addcmul (4 streams)', ms, 4*n*4/ms/1e9,'TB/s')
ms=t(lambda: a.sum()); print('sum   (1 stream)', ms, n*4/ms/1e9,'TB/s')
