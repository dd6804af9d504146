#!/usr/bin/env python3 -B
"""Golden fixture for ``UNet(spatial=3)`` FROM THE REFERENCE ITSELF (sda/nn.py:114-118 picks nn.Conv3d; score.py:66-93).

Same mechanics as make_golden.py (the reference's two files loaded with the zuko stand-in, torch-CPU); kept in its own script so
that the other fixtures' random streams are untouched.  Usage:  python3 -B tests/golden/make_golden_3d.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import _load_reference, _np, _save  # noqa: E402


def main():
    rnn, rscore = _load_reference()
    # (a) two levels, circular, a broadcast context channel, one shared time
    torch.manual_seed(11)
    net = rscore.ScoreUNet(2, context=1, embedding=8, hidden_channels=(4, 8), hidden_blocks=(1, 1), kernel_size=3,
                           activation=torch.nn.SiLU, spatial=3, padding_mode='circular')
    x = torch.randn(2, 2, 4, 8, 6, requires_grad=True)
    c = torch.randn(1, 4, 8, 6)
    t = torch.tensor(0.43)
    cot = torch.randn(2, 2, 4, 8, 6)
    out = net(x, t, c)
    gx, = torch.autograd.grad((out * cot).sum(), x)
    # (b) one level wider, zero padding, ELU, per-sample times, anisotropic kernel
    torch.manual_seed(12)
    netb = rscore.ScoreUNet(3, embedding=8, hidden_channels=(5, 20), hidden_blocks=(1, 2), kernel_size=(1, 3, 3),
                            stride=(1, 2, 2), activation=torch.nn.ELU, spatial=3)
    xb = torch.randn(3, 3, 3, 6, 4, requires_grad=True)
    tb = torch.tensor([0.2, 0.55, 0.9])
    cotb = torch.randn(3, 3, 3, 6, 4)
    outb = netb(xb, tb)
    gxb, = torch.autograd.grad((outb * cotb).sum(), xb)
    _save('unet3d_tiny', sd_a=_np(net.state_dict()), x_a=x, c_a=c, t_a=t, cot_a=cot, out_a=out, gx_a=gx,
          sd_b=_np(netb.state_dict()), x_b=xb, t_b=tb, cot_b=cotb, out_b=outb, gx_b=gxb)


if __name__ == '__main__':
    main()
