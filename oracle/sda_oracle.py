"""CPU restatement (torch, fp32 or fp64) of the reference's posterior-sampling path.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

This is *not* the reference's code: it is a functional restatement that walks a
flat ``state_dict`` (the reference's key names) instead of
building ``nn.Module`` trees.  Each function cites the reference lines whose
arithmetic it restates (paths relative to /root/reference).

Parity status
-------------
* Everything except the channel LayerNorm is pinned against the reference's own
  ``sda/nn.py`` + ``sda/score.py`` executed in the build container
  (``tests/golden/make_golden.py``; fixtures in ``tests/golden/*.npz``).
* The LayerNorm arithmetic lives in the third-party package ``zuko==0.1.4``
  (environment.yml:23), which is NOT in /root/reference and not installable here.
  Its published algorithm is restated in :func:`layer_norm`:
  ``(x - mean) / sqrt(var + eps)`` with the *unbiased* variance and ``eps=1e-5``,
  no affine parameters.  The reference has no tests or golden vectors for it, so
  at that one boundary this oracle is **parity unpinned**; the convention is the
  single switch ``LN_UNBIASED`` below so fixtures can be regenerated in minutes.
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

# zuko 0.1.4 LayerNorm convention (see module docstring): unbiased variance.
LN_UNBIASED = True
LN_EPS = 1e-5

StateDict = Dict[str, Tensor]


# --------------------------------------------------------------------------- #
# activations (sda/utils.py:19-25 names)
# --------------------------------------------------------------------------- #

def activation(name: str) -> Callable[[Tensor], Tensor]:
    table = {
        'ReLU': F.relu,
        'ELU': F.elu,
        'GELU': F.gelu,
        'SELU': F.selu,
        'SiLU': F.silu,
    }
    return table[name]


# --------------------------------------------------------------------------- #
# zuko.nn.LayerNorm (call sites sda/nn.py:61,137,163)
# --------------------------------------------------------------------------- #

def layer_norm(x: Tensor, dim: int = -1, eps: float = LN_EPS, unbiased: Optional[bool] = None) -> Tensor:
    """Standardise along ``dim``; no affine.  zuko 0.1.4 ``nn.LayerNorm.forward``."""
    if unbiased is None:
        unbiased = LN_UNBIASED
    n = x.shape[dim]
    mean = x.mean(dim=dim, keepdim=True)
    cen = x - mean
    var = cen.square().sum(dim=dim, keepdim=True) / (n - 1 if unbiased else n)
    return cen / torch.sqrt(var + eps)


# --------------------------------------------------------------------------- #
# time embedding (sda/score.py:15-35)
# --------------------------------------------------------------------------- #

def time_features(t: Tensor) -> Tensor:
    """[cos(pi j t), sin(pi j t)] for j = 1..16 (score.py:29-33)."""
    freqs = math.pi * torch.arange(1, 17, dtype=t.dtype)
    ang = t.unsqueeze(-1) * freqs
    return torch.cat((torch.cos(ang), torch.sin(ang)), dim=-1)


def time_embedding(sd: StateDict, prefix: str, t: Tensor) -> Tensor:
    """Linear(32,256) -> SiLU -> Linear(256,E) on the features (score.py:22-35)."""
    feats = time_features(t)
    # the reference multiplies by its registered ``freqs`` buffer; honour a loaded one
    key = prefix + 'freqs'
    if key in sd:
        ang = t.unsqueeze(-1) * sd[key].to(t.dtype)
        feats = torch.cat((torch.cos(ang), torch.sin(ang)), dim=-1)
    h = F.linear(feats, sd[prefix + '0.weight'], sd[prefix + '0.bias'])
    h = F.silu(h)
    return F.linear(h, sd[prefix + '2.weight'], sd[prefix + '2.bias'])


# --------------------------------------------------------------------------- #
# U-Net (sda/nn.py:74-206)
# --------------------------------------------------------------------------- #

@dataclass
class UNetConfig:
    in_channels: int
    out_channels: int
    mod_features: int
    hidden_channels: Sequence[int] = (32, 64, 128)
    hidden_blocks: Sequence[int] = (2, 3, 5)
    kernel_size: int = 3
    stride: int = 2
    activation: str = 'ReLU'
    spatial: int = 2
    padding_mode: str = 'zeros'


def _conv(x: Tensor, w: Tensor, b: Tensor, spatial: int, stride: int, padding_mode: str) -> Tensor:
    """nn.ConvNd with padding=k//2 and the given padding mode (nn.py:126-129)."""
    ks = w.shape[2:]
    pads = [k // 2 for k in ks]
    conv = {1: F.conv1d, 2: F.conv2d, 3: F.conv3d}[spatial]            # (nn.py:114-118)
    if padding_mode == 'zeros':
        return conv(x, w, b, stride=stride, padding=pads)
    if padding_mode != 'circular':
        raise NotImplementedError(padding_mode)
    # torch pads last dim first
    flat = []
    for p in reversed(pads):
        flat += [p, p]
    return conv(F.pad(x, flat, mode='circular'), w, b, stride=stride)


def _mod_block(sd: StateDict, p: str, cfg: UNetConfig, x: Tensor, y: Tensor) -> Tensor:
    """x + conv2(act(conv1(LN_c(x + Linear(y)))))   (nn.py:27-28, 131-142)."""
    m = F.linear(y, sd[p + 'project.0.weight'], sd[p + 'project.0.bias'])
    m = m.reshape(m.shape + (1,) * cfg.spatial)
    h = layer_norm(x + m, dim=-(cfg.spatial + 1))
    h = _conv(h, sd[p + 'residue.1.weight'], sd[p + 'residue.1.bias'], cfg.spatial, 1, cfg.padding_mode)
    h = activation(cfg.activation)(h)
    h = _conv(h, sd[p + 'residue.3.weight'], sd[p + 'residue.3.bias'], cfg.spatial, 1, cfg.padding_mode)
    return x + h


def unet_forward(sd: StateDict, prefix: str, cfg: UNetConfig, x: Tensor, y: Tensor,
                 taps: Optional[dict] = None) -> Tensor:
    """UNet.forward (nn.py:184-206).  ``taps`` (optional dict) records intermediates.

    Note the reference stores ``tails`` and ``ascent`` reversed (nn.py:179-182):
    index 0 of either list is the *deepest* level.
    """
    depth = len(cfg.hidden_blocks)
    skips = []
    for lvl in range(depth):
        hp = f'{prefix}heads.{lvl}.' + ('0.' if lvl > 0 else '')
        x = _conv(x, sd[hp + 'weight'], sd[hp + 'bias'], cfg.spatial,
                  cfg.stride if lvl > 0 else 1, cfg.padding_mode)
        if taps is not None:
            taps[f'head{lvl}'] = x
        for b in range(cfg.hidden_blocks[lvl]):
            x = _mod_block(sd, f'{prefix}descent.{lvl}.{b}.', cfg, x, y)
            if taps is not None:
                taps[f'descent{lvl}.{b}'] = x
        skips.append(x)
    skips.pop()
    for j in range(depth):
        lvl = depth - 1 - j
        for b in range(cfg.hidden_blocks[lvl]):
            x = _mod_block(sd, f'{prefix}ascent.{j}.{b}.', cfg, x, y)
            if taps is not None:
                taps[f'ascent{lvl}.{b}'] = x
        if lvl > 0:
            tp = f'{prefix}tails.{j}.2.'
            h = layer_norm(x, dim=-(cfg.spatial + 1))
            strides = (cfg.stride,) * cfg.spatial if isinstance(cfg.stride, int) else tuple(cfg.stride)
            for ax, st in enumerate(strides):               # nn.Upsample(scale_factor=strides, mode='nearest'), nn.py:164
                h = h.repeat_interleave(st, dim=ax - cfg.spatial)
            h = _conv(h, sd[tp + 'weight'], sd[tp + 'bias'], cfg.spatial, 1, cfg.padding_mode)
            x = h + skips.pop()
        else:
            tp = f'{prefix}tails.{j}.'
            x = _conv(x, sd[tp + 'weight'], sd[tp + 'bias'], cfg.spatial, 1, cfg.padding_mode)
        if taps is not None:
            taps[f'tail{lvl}'] = x
    return x


# --------------------------------------------------------------------------- #
# ResMLP / ScoreNet (sda/nn.py:31-71, sda/score.py:38-63)
# --------------------------------------------------------------------------- #

@dataclass
class ResMLPConfig:
    in_features: int
    out_features: int
    hidden_features: Sequence[int] = (64, 64)
    activation: str = 'ReLU'


def resmlp_forward(sd: StateDict, prefix: str, cfg: ResMLPConfig, x: Tensor) -> Tensor:
    """Sequential of [Linear if widths differ, x + Lin(act(Lin(LN(x))))]  (nn.py:50-66)."""
    idx = 0
    widths_in = (cfg.in_features, *cfg.hidden_features)
    widths_out = (*cfg.hidden_features, cfg.out_features)
    act = activation(cfg.activation)
    for before, after in zip(widths_in, widths_out):
        if after != before:
            x = F.linear(x, sd[f'{prefix}{idx}.weight'], sd[f'{prefix}{idx}.bias'])
            idx += 1
        h = layer_norm(x, dim=-1)
        h = F.linear(h, sd[f'{prefix}{idx}.1.weight'], sd[f'{prefix}{idx}.1.bias'])
        h = act(h)
        h = F.linear(h, sd[f'{prefix}{idx}.3.weight'], sd[f'{prefix}{idx}.3.bias'])
        x = x + h
        idx += 1
    return x


def score_net(sd: StateDict, prefix: str, cfg: ResMLPConfig, x: Tensor, t: Tensor,
              c: Optional[Tensor] = None) -> Tensor:
    """ScoreNet.forward: cat(x, emb(t)[, c]) along features -> ResMLP (score.py:53-63)."""
    emb = time_embedding(sd, prefix + 'embedding.', t)
    batch = torch.broadcast_shapes(x.shape[:-1], emb.shape[:-1],
                                   *(() if c is None else (c.shape[:-1],)))
    parts = [x.expand(batch + x.shape[-1:]), emb.expand(batch + emb.shape[-1:])]
    if c is not None:
        parts.append(c.expand(batch + c.shape[-1:]))
    return resmlp_forward(sd, prefix + 'network.', cfg, torch.cat(parts, dim=-1))


# --------------------------------------------------------------------------- #
# ScoreUNet / LocalScoreUNet (sda/score.py:66-93, experiments/kolmogorov/utils.py:29-46)
# --------------------------------------------------------------------------- #

def kolmogorov_forcing(size: int, dtype=torch.float32) -> Tensor:
    """sin(4 * 2pi/size * (i + 1/2)) broadcast to (1, size, size)  (kolmogorov/utils.py:40-41)."""
    domain = 2 * math.pi / size * (torch.arange(size, dtype=dtype) + 0.5)
    return torch.sin(4 * domain).expand(1, size, size).clone()


def score_unet(sd: StateDict, prefix: str, cfg: UNetConfig, x: Tensor, t: Tensor,
               c: Optional[Tensor] = None, taps: Optional[dict] = None) -> Tensor:
    """ScoreUNet.forward (score.py:81-93): concat context channels, flatten batch, embed t."""
    dims = cfg.spatial + 1
    if c is None:
        y = x
    else:
        batch = torch.broadcast_shapes(x.shape[:-dims], c.shape[:-dims])
        y = torch.cat((x.expand(batch + x.shape[-dims:]), c.expand(batch + c.shape[-dims:])), dim=-dims)
    y = y.reshape(-1, *y.shape[-dims:])
    emb = time_embedding(sd, prefix + 'embedding.', t.reshape(-1))
    out = unet_forward(sd, prefix + 'network.', cfg, y, emb, taps=taps)
    return out.reshape(x.shape)


# --------------------------------------------------------------------------- #
# Markov-chain composition (sda/score.py:96-164)
# --------------------------------------------------------------------------- #

def unfold(x: Tensor, order: int) -> Tensor:
    """(B, L, C, ...) -> (B, L-2k, (2k+1)C, ...): window i, slot j holds frame i+j (score.py:146-153)."""
    w = 2 * order + 1
    n = x.shape[1] - 2 * order
    frames = [x[:, j:j + n] for j in range(w)]       # each (B, n, C, ...)
    return torch.cat(frames, dim=2)


def fold(s: Tensor, order: int) -> Tensor:
    """Selective gather back to (B, L, C, ...)  (score.py:155-164).

    First window gives its slots 0..k-1, every window gives its centre slot k,
    last window gives its slots k+1..2k.  No averaging.
    """
    w = 2 * order + 1
    B, n = s.shape[:2]
    C = s.shape[2] // w
    s = s.reshape(B, n, w, C, *s.shape[3:])
    return torch.cat((s[:, 0, :order], s[:, :, order], s[:, -1, w - order:]), dim=1)


def fold_index_map(L: int, order: int) -> Tuple[list, list]:
    """(source window, source slot) for every output position l in 0..L-1."""
    n = L - 2 * order
    win, slot = [], []
    for l in range(L):
        if l < order:
            win.append(0); slot.append(l)
        elif l >= L - order:
            win.append(n - 1); slot.append(l - (n - 1))
        else:
            win.append(l - order); slot.append(order)
    return win, slot


def mc_score_net(kernel: Callable[[Tensor, Tensor, Optional[Tensor]], Tensor], order: int,
                 x: Tensor, t: Tensor, c: Optional[Tensor] = None) -> Tensor:
    """MCScoreNet.forward (score.py:134-144)."""
    return fold(kernel(unfold(x, order), t, c), order)


def mc_score_wrapper(score: Callable[[Tensor, Tensor, Optional[Tensor]], Tensor],
                     x: Tensor, t: Tensor, c: Optional[Tensor] = None) -> Tensor:
    """MCScoreWrapper.forward (score.py:104-110): (B,L,C,...) <-> (B,C,L,...)."""
    return score(x.transpose(1, 2), t, c).transpose(1, 2)


# --------------------------------------------------------------------------- #
# VP-SDE schedule and predictor-corrector sampler (sda/score.py:167-263, 279-302)
# --------------------------------------------------------------------------- #

@dataclass
class Schedule:
    alpha: str = 'cos'
    eta: float = 1e-3
    kind: str = 'vp'          # 'vp' | 'subvp' | 'subsubvp'

    def mu(self, t: Tensor) -> Tensor:
        if self.alpha == 'lin':
            return 1 - (1 - self.eta) * t
        if self.alpha == 'cos':
            return torch.cos(math.acos(math.sqrt(self.eta)) * t) ** 2
        if self.alpha == 'exp':
            return torch.exp(math.log(self.eta) * t ** 2)
        raise ValueError(self.alpha)

    def sigma(self, t: Tensor) -> Tensor:
        a = self.mu(t)
        if self.kind == 'vp':
            return (1 - a ** 2 + self.eta ** 2).sqrt()        # score.py:209-210
        if self.kind == 'subvp':
            return 1 - a ** 2 + self.eta                      # score.py:287-288
        if self.kind == 'subsubvp':
            return 1 - a + self.eta                           # score.py:300-301
        raise ValueError(self.kind)


def sample(eps: Callable[[Tensor, Tensor], Tensor], sched: Schedule, x1: Tensor, event_ndim: int,
           steps: int = 64, corrections: int = 0, tau: float = 1.0,
           noise: Optional[Callable[[int, int], Tensor]] = None,
           record: Optional[list] = None) -> Tensor:
    """VPSDE.sample (score.py:225-263) from a given initial draw ``x1`` of shape (N, *event).

    ``noise(step, corr)`` supplies the corrector draws (injected for parity); default randn_like.
    ``record`` (optional list) receives ``x`` after every step.
    """
    x = x1
    dims = tuple(range(-event_ndim, 0))
    time = torch.linspace(1, 0, steps + 1, dtype=torch.float32).to(x.dtype)
    dt = 1 / steps
    with torch.no_grad():
        for i, t in enumerate(time[:-1]):
            r = sched.mu(t - dt) / sched.mu(t)
            x = r * x + (sched.sigma(t - dt) - r * sched.sigma(t)) * eps(x, t)
            for j in range(corrections):
                z = torch.randn_like(x) if noise is None else noise(i, j)
                e = eps(x, t - dt)
                delta = tau / e.square().mean(dim=dims, keepdim=True)
                x = x - (delta * e + torch.sqrt(2 * delta) * z) * sched.sigma(t - dt)
            if record is not None:
                record.append(x.clone())
    return x


# --------------------------------------------------------------------------- #
# likelihood guidance (sda/score.py:305-396)
# --------------------------------------------------------------------------- #

def gaussian_score(eps: Callable[[Tensor, Tensor], Tensor], sched: Schedule, y: Tensor,
                   A: Callable[[Tensor], Tensor], std, gamma, x: Tensor, t: Tensor,
                   detach: bool = False, return_grad: bool = False):
    """GaussianScore.forward (score.py:375-396)."""
    mu, sigma = sched.mu(t), sched.sigma(t)
    std = torch.as_tensor(std, dtype=x.dtype)
    gamma = torch.as_tensor(gamma, dtype=x.dtype)
    if detach:
        with torch.no_grad():
            e = eps(x, t)
    with torch.enable_grad():
        xg = x.detach().requires_grad_(True)
        if not detach:
            e = eps(xg, t)
        x_hat = (xg - sigma * e) / mu
        err = y - A(x_hat)
        var = std ** 2 + gamma * (sigma / mu) ** 2
        log_p = -(err ** 2 / var).sum() / 2
    s, = torch.autograd.grad(log_p, xg)
    out = (e - sigma * s).detach()
    return (out, s.detach()) if return_grad else out


def dps_gaussian_score(eps: Callable[[Tensor, Tensor], Tensor], sched: Schedule, y: Tensor,
                       A: Callable[[Tensor], Tensor], zeta: float, x: Tensor, t: Tensor) -> Tensor:
    """DPSGaussianScore.forward (score.py:331-344).  ``err`` couples the whole batch."""
    mu, sigma = sched.mu(t), sched.sigma(t)
    with torch.enable_grad():
        xg = x.detach().requires_grad_(True)
        e = eps(xg, t)
        x_hat = (xg - sigma * e) / mu
        err = (y - A(x_hat)).square().sum()
    s, = torch.autograd.grad(err, xg)
    s = -s * zeta / err.sqrt()
    return (e - sigma * s).detach()


# --------------------------------------------------------------------------- #
# observation operators used by the reference's notebooks (sda/mcs.py:340-347, 361-375)
# --------------------------------------------------------------------------- #

def coarsen(x: Tensor, r: int = 2) -> Tensor:
    """Block mean over r x r cells (mcs.py:340-347)."""
    *batch, h, w = x.shape
    return x.reshape(*batch, h // r, r, w // r, r).mean(dim=(-3, -1))


def vorticity(x: Tensor) -> Tensor:
    """d u/d x - d v/d y by central differences on the periodic grid (mcs.py:361-375)."""
    *batch, _, h, w = x.shape
    y = x.reshape(-1, 2, h, w)
    du = (torch.roll(y[:, 0], -1, dims=-1) - torch.roll(y[:, 0], 1, dims=-1)) / 2
    dv = (torch.roll(y[:, 1], -1, dims=-2) - torch.roll(y[:, 1], 1, dims=-2)) / 2
    return (du - dv).reshape(*batch, h, w)


# --------------------------------------------------------------------------- #
# random-init state dicts with the reference's key names and init distributions
# --------------------------------------------------------------------------- #

def _init_linear(gen: torch.Generator, out_f: int, in_f: int, dtype) -> Tuple[Tensor, Tensor]:
    bound = 1 / math.sqrt(in_f)
    w = (torch.rand(out_f, in_f, generator=gen, dtype=dtype) * 2 - 1) * bound
    b = (torch.rand(out_f, generator=gen, dtype=dtype) * 2 - 1) * bound
    return w, b


def _init_conv(gen: torch.Generator, out_c: int, in_c: int, ks: Sequence[int], dtype) -> Tuple[Tensor, Tensor]:
    fan_in = in_c * math.prod(ks)
    bound = 1 / math.sqrt(fan_in)
    w = (torch.rand(out_c, in_c, *ks, generator=gen, dtype=dtype) * 2 - 1) * bound
    b = (torch.rand(out_c, generator=gen, dtype=dtype) * 2 - 1) * bound
    return w, b


def init_time_embedding(gen, prefix: str, features: int, dtype=torch.float32) -> StateDict:
    sd = {prefix + 'freqs': math.pi * torch.arange(1, 17, dtype=dtype)}
    sd[prefix + '0.weight'], sd[prefix + '0.bias'] = _init_linear(gen, 256, 32, dtype)
    sd[prefix + '2.weight'], sd[prefix + '2.bias'] = _init_linear(gen, features, 256, dtype)
    return sd


def init_unet(gen, prefix: str, cfg: UNetConfig, dtype=torch.float32) -> StateDict:
    """Same key set / shapes as the reference UNet's state_dict (uniform +-1/sqrt(fan_in) init)."""
    sd: StateDict = {}
    ks = [cfg.kernel_size] * cfg.spatial
    depth = len(cfg.hidden_blocks)
    ch = list(cfg.hidden_channels)

    def block(p, c):
        sd[p + 'project.0.weight'], sd[p + 'project.0.bias'] = _init_linear(gen, c, cfg.mod_features, dtype)
        sd[p + 'residue.1.weight'], sd[p + 'residue.1.bias'] = _init_conv(gen, c, c, ks, dtype)
        sd[p + 'residue.3.weight'], sd[p + 'residue.3.bias'] = _init_conv(gen, c, c, ks, dtype)

    for lvl in range(depth):
        j = depth - 1 - lvl
        if lvl == 0:
            sd[f'{prefix}heads.0.weight'], sd[f'{prefix}heads.0.bias'] = _init_conv(gen, ch[0], cfg.in_channels, ks, dtype)
            sd[f'{prefix}tails.{j}.weight'], sd[f'{prefix}tails.{j}.bias'] = _init_conv(gen, cfg.out_channels, ch[0], ks, dtype)
        else:
            sd[f'{prefix}heads.{lvl}.0.weight'], sd[f'{prefix}heads.{lvl}.0.bias'] = _init_conv(gen, ch[lvl], ch[lvl - 1], ks, dtype)
            sd[f'{prefix}tails.{j}.2.weight'], sd[f'{prefix}tails.{j}.2.bias'] = _init_conv(gen, ch[lvl - 1], ch[lvl], ks, dtype)
        for b in range(cfg.hidden_blocks[lvl]):
            block(f'{prefix}descent.{lvl}.{b}.', ch[lvl])
            block(f'{prefix}ascent.{j}.{b}.', ch[lvl])
    return sd


def init_score_unet(seed: int, prefix: str, cfg: UNetConfig, dtype=torch.float32) -> StateDict:
    gen = torch.Generator().manual_seed(seed)
    sd = init_time_embedding(gen, prefix + 'embedding.', cfg.mod_features, dtype)
    sd.update(init_unet(gen, prefix + 'network.', cfg, dtype))
    return sd


def init_score_net(seed: int, prefix: str, cfg: ResMLPConfig, embedding: int, dtype=torch.float32) -> StateDict:
    gen = torch.Generator().manual_seed(seed)
    sd = init_time_embedding(gen, prefix + 'embedding.', embedding, dtype)
    idx = 0
    for before, after in zip((cfg.in_features, *cfg.hidden_features), (*cfg.hidden_features, cfg.out_features)):
        if after != before:
            sd[f'{prefix}network.{idx}.weight'], sd[f'{prefix}network.{idx}.bias'] = _init_linear(gen, after, before, dtype)
            idx += 1
        sd[f'{prefix}network.{idx}.1.weight'], sd[f'{prefix}network.{idx}.1.bias'] = _init_linear(gen, after, after, dtype)
        sd[f'{prefix}network.{idx}.3.weight'], sd[f'{prefix}network.{idx}.3.bias'] = _init_linear(gen, after, after, dtype)
        idx += 1
    return sd


def cast_sd(sd: StateDict, dtype) -> StateDict:
    return {k: v.to(dtype) for k, v in sd.items()}


# ---------------------------------------------------------------------------------------------- evaluation metrics
# (test infrastructure like everything else in this file)

MMD_BANDWIDTHS = (1e-3, 1e-2, 1e-1, 1e-0, 1e1, 1e2, 1e3)


def mmd(x: Tensor, y: Tensor, exact: bool = False) -> Tensor:
    """sda/utils.py:222-263.  ``exact=False`` follows the reference's arithmetic (Gram matrices, squared distances as
    |x|^2 + |y|^2 - 2 x.y in the input dtype); ``exact=True`` forms the differences directly in float64 -- the value the
    reference's formula approximates, free of its fp32 cancellation on the small bandwidths.  Pinned against
    tests/golden/metrics_mmd.npz (the reference's own output)."""
    x, y = x.flatten(1), y.flatten(1)
    if exact:
        x, y = x.double(), y.double()
        err_xx = (x[:, None] - x[None]).square().sum(-1)
        err_yy = (y[:, None] - y[None]).square().sum(-1)
        err_xy = (x[:, None] - y[None]).square().sum(-1)
    else:
        xx, yy, xy = x @ x.T, y @ y.T, x @ y.T
        dxx, dyy = xx.diag().unsqueeze(1), yy.diag().unsqueeze(0)
        err_xx = dxx + dxx.T - 2 * xx
        err_yy = dyy + dyy.T - 2 * yy
        err_xy = dxx + dyy - 2 * xy
    total = 0
    for sigma in MMD_BANDWIDTHS:
        total = total + torch.exp(-err_xx / sigma).mean() + torch.exp(-err_yy / sigma).mean() \
            - 2 * torch.exp(-err_xy / sigma).mean()
    return total


def bpf(x: Tensor, y: Tensor, transition: Callable[[Tensor], Tensor], likelihood: Callable[[Tensor, Tensor], Tensor],
        step: int = 1) -> Tensor:
    """sda/utils.py:168-200: propagate, weight, multinomial resampling of whole histories.  Pinned against
    tests/golden/metrics_bpf.npz."""
    hist = [x]
    for yi in y:
        for _ in range(step):
            hist.append(transition(hist[-1]))
        w = likelihood(yi, hist[-1])
        j = torch.multinomial(w, len(w), replacement=True)
        hist = [h[j] for h in hist]
    return torch.stack(hist, dim=1)


def emd(x: Tensor, y: Tensor) -> Tensor:
    """sda/utils.py:203-219: ``ot.emd2([], [], cdist(x, y))`` -- POT (PyPI ``POT``, unpinned in the reference's
    environment file) solves min_P <P, C> over couplings with uniform marginals.  PARITY UNPINNED: POT is absent here, so
    no fixture of the reference's own output exists.  Restated from the LP's definition: for equally many samples a
    vertex of the transport polytope is a permutation / n (Birkhoff), so the optimum is a linear assignment (scipy's
    solver); for unequal counts the LP itself is solved (scipy.optimize.linprog, small cases only)."""
    import numpy as np
    from scipy.optimize import linear_sum_assignment, linprog
    c = torch.cdist(x.flatten(1).double(), y.flatten(1).double()).numpy()
    m, n = c.shape
    if m == n:
        r, cidx = linear_sum_assignment(c)
        return x.new_tensor(c[r, cidx].sum() / n)
    a_eq = np.zeros((m + n, m * n))
    for i in range(m):
        a_eq[i, i * n:(i + 1) * n] = 1
    for j in range(n):
        a_eq[m + j, j::n] = 1
    b_eq = np.concatenate([np.full(m, 1 / m), np.full(n, 1 / n)])
    res = linprog(c.reshape(-1), A_eq=a_eq, b_eq=b_eq, bounds=(0, None), method='highs')
    return x.new_tensor(res.fun)
