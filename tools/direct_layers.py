#!/usr/bin/env python3
"""Which convolution launches of a guided Kolmogorov step take the direct kernel, and what they cost (GPU box):
    python tools/direct_layers.py [size]
One guided score evaluation (forward + VJP) of the reference Kolmogorov net on `size` x `size` windows; every launch that is
not served by conv_wino4 is listed by shape with its summed HIP-event time."""
import collections
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sda_amd import _lib, ops  # noqa: E402
from sda_amd import observe as Ob  # noqa: E402
from sda_amd.experiments.kolmogorov import make_score  # noqa: E402
from sda_amd.score import GaussianScore, VPSDE  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = make_score(window=5, embedding=64, hidden_channels=(96, 192, 384), hidden_blocks=(3, 3, 3), kernel_size=3, activation='SiLU',
                 size=size).to(dev)
A = Ob.Subsample((slice(None, None, 4), slice(None), slice(None, None, 4), slice(None, None, 4)))
L = 8
y = torch.randn_like(A(torch.empty(1, L, 2, size, size, device=dev)))
gs = GaussianScore(y, A=A, std=0.1, sde=VPSDE(net, shape=()), gamma=1e-2).to(dev)
x = torch.randn(1, L, 2, size, size, device=dev)
t = torch.tensor(0.5, device=dev)
records = []
orig = ops.conv_igemm


def spy(desc):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(_lib.load().sda_conv_igemm(ctypes.byref(desc), ops._stream()), 'sda_conv_igemm')
    e1.record()
    key = (ops.conv_path(desc), desc.cx + desc.cctx, desc.cout, desc.kh, desc.kw, desc.stride_h, desc.zins_h, desc.up_h, desc.ho, desc.wo,
           bool(desc.res), bool(desc.dact_z), bool(desc.ln_mean), bool(desc.mod), desc.act_in)
    records.append((key, e0, e1, desc.n))


ops.conv_igemm = spy
import sda_amd.engine as E  # noqa: E402
if hasattr(E, 'ops'):
    E.ops.conv_igemm = spy
for _ in range(2):
    records.clear()
    gs(x, t)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, a, b, n in records:
    r = agg.setdefault(key, [0, 0.0, 0])
    r[0] += 1
    r[1] += a.elapsed_time(b)
    r[2] += n
tot = sum(r[1] for r in agg.values())
print('path cin cout kh kw stride zins up ho wo res dact ln mod act | launches images ms share')
for key, r in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(*key, '|', r[0], r[2], f'{r[1]:.2f}', f'{100 * r[1] / tot:.1f}%')
