import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sda_amd import ops
dev = torch.device('cuda:0')
SHAPES = ((96, 256, int(os.environ.get("LN_N", "16"))), (192, 128, int(os.environ.get("LN_N", "16"))), (384, 64, int(os.environ.get("LN_N", "16"))), (96, 64, 128))
if os.environ.get('LN_BM64'):           # the reference's default widths (64, 128, 256) at the kolmogorov64_default sizes
    SHAPES = ((64, 64, 960), (128, 32, 960), (256, 16, 960))
for c, hw_, n in SHAPES:
    h = w = hw_
    x = torch.randn(n, c, h, w, device=dev); gh = torch.randn_like(x); res = torch.randn_like(x)
    mod = torch.randn(1, c, device=dev)
    mean = torch.empty(n * h * w, device=dev); rstd = torch.empty_like(mean)
    ops.ln_stats(x, mod, 0, 1e-5, True, mean, rstd)
    gx = torch.empty_like(x)
    def runs():
        ops.ln_stats(x, mod, 0, 1e-5, True, mean, rstd)
    for _ in range(5): runs()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): runs()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f'ln_stats c={c} {h}x{w} n={n}: {ms:.3f} ms  {4.0 * n * h * w * (c + 2) / ms / 1e9:.2f} TB/s  quad={os.environ.get("SDA_LN_STATS_QUAD", "1")}')
    xm0 = (x + mod[:, :, None, None]).double(); v0, m0 = torch.var_mean(xm0, dim=1, unbiased=True)
    print('   stats rel err', ((mean.double().reshape(m0.shape) - m0).abs().max() / m0.abs().max()).item(), ((rstd.double().reshape(v0.shape) - 1 / torch.sqrt(v0 + 1e-5)).abs().max() / (1 / torch.sqrt(v0 + 1e-5)).abs().max()).item())
    def run():
        ops.ln_bwd(gh, x, h, w, mod, 0, mean, rstd, True, (1, 1), res, gx)
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    nb = 4.0 * n * h * w * (c * 4 + 2)
    print(f'c={c} {h}x{w} n={n}: {ms:.3f} ms  {nb / ms / 1e9:.2f} TB/s  quad={os.environ.get("SDA_LN_BWD_QUAD", "1")}')
    # reference check
    xm = (x + mod[:, :, None, None]).double()
    xm.requires_grad_(True)
    var, mu = torch.var_mean(xm, dim=1, unbiased=True, keepdim=True)
    y = (xm - mu) / torch.sqrt(var + 1e-5)
    ref, = torch.autograd.grad(y, xm, gh.double())
    ref = ref + res.double()
    print('   rel err', ((gx.double() - ref).abs().max() / ref.abs().max()).item())
