"""Layer timings of the general 3-D convolution kernel (csrc/conv3d.hip) on K64-like widths; prints TFLOP/s (2 x MACs) per launch type.
    /usr/local/graft/bin/gpurun --timeout 600 -- 'python tools/conv3d_bench.py'"""
import sys

import torch

sys.path.insert(0, '.')
from sda_amd.engine3d import _Conv3d  # noqa: E402


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    for cin, cout, size, n, stride, up in ((96, 96, 32, 8, 1, 1), (192, 192, 16, 8, 1, 1), (384, 384, 8, 8, 1, 1),
                                           (96, 192, 32, 8, 2, 1), (192, 96, 16, 8, 1, 2), (11, 96, 32, 8, 1, 1), (96, 10, 32, 8, 1, 1)):
        conv = torch.nn.Conv3d(cin, cout, 3, stride=stride, padding=1, padding_mode='circular').to(dev)
        op = _Conv3d(conv)
        x = torch.randn(n, cin, size, size, size, device=dev)
        out = op.forward(x, up=(up,) * 3)
        flops = 2.0 * 27 * cin * cout * out[0, 0].numel() * n
        ms = timed(lambda: op.forward(x, up=(up,) * 3))
        g = torch.randn_like(out)
        insz = tuple(s * up for s in x.shape[2:])
        msb = timed(lambda: op.vjp(g, insz))
        print(f'conv3d {cin:4d}->{cout:4d} {size}^3 x{n} stride {stride} up {up}: fwd {ms:7.3f} ms {flops / ms / 1e9:6.1f} TFLOP/s | '
              f'vjp {msb:7.3f} ms {flops / msb / 1e9:6.1f} TFLOP/s', flush=True)


if __name__ == '__main__':
    main()
