"""TEST-ONLY host shim: lets the *Python orchestration* of sda_amd (engine forward / VJP sequencing, autograd
Functions, guidance chain rule, the PC loop) execute on CPU tensors so it can be checked against the golden
fixtures without a GPU.

The shim monkeypatches `sda_amd.ops` inside a test:
  * convolutions go through libsda_emu.so -- the host replay of the gfx950 conv tile algorithm (same planner and
    index helpers as the device kernel);
  * the small streaming kernels are replaced by a few lines of torch each.
Nothing here is reachable from the product: sda_amd itself has no CPU path and raises on CPU tensors.
"""
import ctypes

import torch

from sda_amd import build as sbuild
from sda_amd import engine as E
from sda_amd import ops
from sda_amd._lib import ConvDesc

_ACT = {1: torch.nn.functional.silu, 2: torch.relu, 3: torch.nn.functional.elu, 4: torch.nn.functional.gelu,
        5: torch.nn.functional.selu}


def install(monkeypatch):
    emu = ctypes.CDLL(sbuild.build_emu())
    emu.sda_conv_igemm_emulate.restype = ctypes.c_int
    emu.sda_conv_igemm_emulate.argtypes = [ctypes.POINTER(ConvDesc)]
    emu.sda_pack_conv_weight_host.restype = None

    def _dev(*ts):
        for t in ts:
            assert t is None or t.dtype == torch.float32

    def conv_igemm(desc):
        rc = emu.sda_conv_igemm_emulate(ctypes.byref(desc))
        assert rc == 0, f'emulator rc={rc}'

    def pack(w, cout, cin, kh, kw, transpose, keep, dst, k_pad, m_pad):
        emu.sda_pack_conv_weight_host(ctypes.c_void_p(w.data_ptr()), cout, cin, kh, kw, int(transpose), keep,
                                      ctypes.c_void_p(dst.data_ptr()), k_pad, m_pad)

    def _u(x, mod, mod_sn):
        n, c = x.shape[:2]
        xv = x.reshape(n, c, -1)
        if mod is None:
            return xv
        # mod is a (possibly offset / strided) view: rows of `mod_sn` floats (or one shared row)
        rows = []
        base = mod.reshape(-1) if mod.is_contiguous() else None
        for i in range(n):
            r = mod[i if mod_sn else 0]
            rows.append(r[:c])
        return xv + torch.stack(rows)[:, :, None]

    def ln_stats(x, mod, mod_sn, eps, unbiased, mean, rstd):
        u = _u(x, mod, mod_sn)
        var, m = torch.var_mean(u, dim=1, unbiased=bool(unbiased))
        mean.copy_(m.reshape(-1))
        rstd.copy_((1 / torch.sqrt(var + eps)).reshape(-1))

    def ln_apply(x, mod, mod_sn, mean, rstd, y):
        u = _u(x, mod, mod_sn)
        n = u.shape[0]
        y.copy_(((u - mean.reshape(n, 1, -1)) * rstd.reshape(n, 1, -1)).reshape(y.shape))

    def ln_bwd(gh, x, h, w, mod, mod_sn, mean, rstd, unbiased, pool, res, gx, out_amax=None):
        n, c = x.shape[:2]
        u = _u(x, mod, mod_sn)
        hh = (u - mean.reshape(n, 1, -1)) * rstd.reshape(n, 1, -1)
        g = gh
        if tuple(pool) != (1, 1):
            g = gh.reshape(n, c, h, pool[0], w, pool[1]).sum((-1, -3))
        g = g.reshape(n, c, -1)
        a = g.mean(1, keepdim=True)
        b = (g * hh).sum(1, keepdim=True) / (c - 1 if unbiased else c)
        out = rstd.reshape(n, 1, -1) * (g - a - hh * b)
        if res is not None:
            out = out + res.reshape(n, c, -1)
        gx.copy_(out.reshape(gx.shape))
        if out_amax is not None:
            out_amax.fill_(float(gx.abs().max()))

    def time_embed(t, freqs, w0, b0, w2, b2):
        ang = t.reshape(-1, 1) * freqs
        f = torch.cat((ang.cos(), ang.sin()), -1)
        return torch.nn.functional.linear(torch.nn.functional.silu(torch.nn.functional.linear(f, w0, b0)), w2, b2)

    def linear_small(x, w, b):
        return torch.nn.functional.linear(x, w, b)

    from oracle import sda_oracle as O

    def fold(s, b, nw, k, c, hw, out):
        out.copy_(O.fold(s.reshape(b, nw, (2 * k + 1) * c, hw), k).reshape(out.shape))

    def fold_adjoint(g_out, b, nw, k, c, hw, g_s):
        with torch.enable_grad():
            s = torch.zeros(b, nw, (2 * k + 1) * c, hw, requires_grad=True)
            gs, = torch.autograd.grad(O.fold(s, k), s, g_out.reshape(b, nw + 2 * k, c, hw))
        g_s.copy_(gs.reshape(g_s.shape))

    def unfold_adjoint(g_win, b, nw, k, c, hw, tot, g_x):
        with torch.enable_grad():
            x = torch.zeros(b, nw + 2 * k, c, hw, requires_grad=True)
            gx, = torch.autograd.grad(O.unfold(x, k), x, g_win.reshape(b, nw, tot, hw)[:, :, :(2 * k + 1) * c])
        g_x.copy_(gx.reshape(g_x.shape))

    def pc_predict(x, eps, r, c1, coef_dev=None):
        x.copy_(r * x + c1 * eps)

    def sumsq_partial(eps, b, partial):
        partial.zero_()
        partial.reshape(b, -1)[:, 0] = eps.reshape(b, -1).square().sum(1)
        return partial.numel() // b

    def pc_correct(x, eps, z, b, partial, tau, sigma, coef_dev=None, nchunk=None):
        per = x.numel() // b
        delta = (tau / (partial.reshape(b, -1).sum(1) / per)).reshape(b, *([1] * (x.dim() - 1)))
        x.copy_(x - (delta * eps + torch.sqrt(2 * delta) * z) * sigma)

    def denoise(x, eps, mu, sigma, xhat):
        xhat.copy_((x - sigma * eps) / mu)

    def guided_combine(eps, ghat, vjp, mu, sigma, out):
        v = 0 if vjp is None else sigma * vjp
        out.copy_(eps - (sigma / mu) * (ghat - v))

    def linear(x, w, b, *, trans_w=False, act_in=0, act_out=0, dact_z=None, act_d=0, res=None):
        xin = _ACT[act_in](x) if act_in else x
        y = xin @ (w if trans_w else w.t())
        if b is not None:
            y = y + b
        if act_out:
            y = _ACT[act_out](y)
        if dact_z is not None:
            with torch.enable_grad():
                zz = dact_z.detach().clone().requires_grad_(True)
                dz, = torch.autograd.grad(_ACT[act_d](zz).sum(), zz)
            y = y * dz
        if res is not None:
            y = y + res
        return y.contiguous()

    def row_ln(x, eps, unbiased, y, mean=None, rstd=None):
        var, m = torch.var_mean(x, dim=-1, unbiased=bool(unbiased), keepdim=True)
        r = 1 / torch.sqrt(var + eps)
        y.copy_((x - m) * r)
        if mean is not None:
            mean.copy_(m.reshape(-1))
        if rstd is not None:
            rstd.copy_(r.reshape(-1))

    def row_ln_bwd(gh, x, mean, rstd, unbiased, res, gx):
        f = x.shape[-1]
        h = (x - mean[:, None]) * rstd[:, None]
        a = gh.mean(-1, keepdim=True)
        b2 = (gh * h).sum(-1, keepdim=True) / (f - 1 if unbiased else f)
        out = rstd[:, None] * (gh - a - h * b2)
        if res is not None:
            out = out + res
        gx.copy_(out)

    def randn_rows(out, seed, row0, draw=0, draw_dev=None, draw_mul=1, draw_add=0):
        from tests import philox_ref
        if draw_dev is not None:
            draw = int(draw_dev.item()) * draw_mul + draw_add
        rows = out.shape[0]
        z = philox_ref.randn_rows(rows, out.numel() // max(rows, 1), seed, row0, draw)
        out.copy_(torch.from_numpy(z).reshape(out.shape))
        return out

    def gauss_cotangent(y, ax, std, gamma, mu, sigma):
        return (y - ax) / (std ** 2 + gamma * (torch.as_tensor(sigma) / torch.as_tensor(mu)) ** 2)

    for name, fn in dict(_dev=_dev, randn_rows=randn_rows, gauss_cotangent=gauss_cotangent, conv_igemm=conv_igemm, pack_conv_weight=pack, ln_stats=ln_stats, ln_apply=ln_apply,
                         ln_bwd=ln_bwd, time_embed=time_embed, linear_small=linear_small, fold=fold,
                         fold_adjoint=fold_adjoint, unfold_adjoint=unfold_adjoint, pc_predict=pc_predict,
                         sumsq_partial=sumsq_partial, pc_correct=pc_correct, denoise=denoise,
                         guided_combine=guided_combine, linear=linear, row_ln=row_ln, row_ln_bwd=row_ln_bwd).items():
        monkeypatch.setattr(ops, name, fn)
    monkeypatch.setattr(ops, 'WINOGRAD', False)      # the Winograd form has no host replay; the direct form is emulated
    monkeypatch.setattr(ops, 'PARITY4', False)       # (the one-launch parity kernel is a device kernel: the four class launches are emulated)
    monkeypatch.setattr(ops, 'NET1D', False)         # (likewise the whole-net 1-D kernel)
    monkeypatch.setattr(ops, 'POOLED', False)        # (and the pooled-output form of the tails' VJP: plain launch + pooling reader)
    monkeypatch.setattr(ops, 'BLOCK1D', False)       # the fused 1-D block is a device kernel: the host replay takes the per-layer path
    monkeypatch.setattr(E.UNetEngine, "chunk_size", lambda self, n, hs, ws, save, device, fraction=None: n)
    _install_3d(monkeypatch)


def _install_3d(monkeypatch):
    """The 3-D engine (sda_amd/engine3d.py) on CPU: its two launch types replaced by torch's conv3d, so that the sequencing of
    heads / blocks / tails and of the hand-written VJP is checked without a GPU (the kernel itself: tests/test_gpu_unet3d.py)."""
    import torch.nn.functional as F
    from sda_amd import engine3d as E3

    def _conv(self, x, up):
        w, pads = self.conv.weight.detach(), list(self.pad)
        for ax, u in enumerate(up):
            x = x.repeat_interleave(u, dim=2 + ax)
        if self.circular:
            flat = []
            for p in reversed(pads):
                flat += [p, p]
            return F.conv3d(F.pad(x, flat, mode='circular'), w, None, stride=self.stride)
        return F.conv3d(x, w, None, stride=self.stride, padding=pads)

    def forward(self, x, *, up=(1, 1, 1), act_in=0, res=None):
        if act_in:
            x = _ACT[act_in](x)
        out = _conv(self, x, up)
        if self.conv.bias is not None:
            out = out + self.conv.bias.detach().reshape(1, -1, 1, 1, 1)
        return (out if res is None else out + res).contiguous()

    def vjp(self, g, in_size, *, act=0, z=None, res=None):
        with torch.enable_grad():
            xin = torch.zeros((g.shape[0], self.cin) + tuple(in_size), requires_grad=True)
            y = _conv(self, xin, (1, 1, 1))
        out, = torch.autograd.grad(y, xin, g)
        if z is not None:
            with torch.enable_grad():
                zz = z.detach().clone().requires_grad_(True)
                dz, = torch.autograd.grad(_ACT[act](zz).sum(), zz)
            out = out * dz
        return (out if res is None else out + res).contiguous()

    def pool_sum(g, f):
        n, c, d, h, w = g.shape
        return g.reshape(n, c, d // f[0], f[0], h // f[1], f[1], w // f[2], f[2]).sum((3, 5, 7)).contiguous()

    monkeypatch.setattr(E3._Conv3d, 'forward', forward)
    monkeypatch.setattr(E3._Conv3d, 'vjp', vjp)
    monkeypatch.setattr(E3, '_pool_sum', pool_sum)
