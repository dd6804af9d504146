#!/usr/bin/env python3 -B
"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference); the fixtures it writes
(`*.npz`, data only) are committed and travel to the GPU box, this script's
imports of the reference do not.

How the reference is executed
-----------------------------
`import sda` fails here (sda/__init__.py eagerly imports jax/h5py/POT users), and
`sda/nn.py` / `sda/score.py` need two symbols from `zuko==0.1.4`, which is absent
and not installable (no network).  So the two hot-path files are loaded directly
with importlib, with a minimal module object standing in for the two zuko
symbols:

* ``zuko.utils.broadcast`` -- shape-only, no arithmetic.
* ``zuko.nn.LayerNorm``   -- restated from zuko 0.1.4's published source:
  ``var_mean(unbiased=True)``, ``(x-mean)/sqrt(var+eps)``, eps=1e-5, no affine.
  THIS IS THE ONE PLACE THE FIXTURES ARE NOT PINNED BY THE REFERENCE'S OWN CODE
  ("parity unpinned" at the zuko boundary).  Flip
  ``LN_UNBIASED`` here and in oracle/sda_oracle.py and re-run to regenerate.

Everything else in every fixture (conv/linear/activation ordering, skip wiring,
unfold/fold, schedule, PC loop, guidance autograd) is the reference's own code
running on torch-CPU.

Usage:  python3 -B tests/golden/make_golden.py
"""

import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
LN_UNBIASED = True

sys.dont_write_bytecode = True
torch.set_num_threads(8)


# ------------------------------------------------------------------ zuko stand-in
def _install_zuko():
    zuko = types.ModuleType('zuko')
    znn = types.ModuleType('zuko.nn')
    zutils = types.ModuleType('zuko.utils')

    class LayerNorm(torch.nn.Module):
        def __init__(self, dim=-1, eps=1e-5):
            super().__init__()
            self.dim = dim if type(dim) is int else tuple(dim)
            self.eps = eps

        def forward(self, x):
            var, mean = torch.var_mean(x, unbiased=LN_UNBIASED, dim=self.dim, keepdim=True)
            return (x - mean) / (var + self.eps).sqrt()

    def broadcast(*tensors, ignore=0):
        if type(ignore) is int:
            ignore = [ignore] * len(tensors)
        dims = [t.dim() - i for t, i in zip(tensors, ignore)]
        common = torch.broadcast_shapes(*(t.shape[:d] for t, d in zip(tensors, dims)))
        return [torch.broadcast_to(t, common + t.shape[d:]) for t, d in zip(tensors, dims)]

    znn.LayerNorm = LayerNorm
    zutils.broadcast = broadcast
    zuko.nn, zuko.utils = znn, zutils
    sys.modules.update({'zuko': zuko, 'zuko.nn': znn, 'zuko.utils': zutils})


def _load_reference():
    _install_zuko()
    pkg = types.ModuleType('sda')
    pkg.__path__ = [os.path.join(REF, 'sda')]
    sys.modules['sda'] = pkg
    mods = {}
    for name in ('nn', 'score'):
        spec = importlib.util.spec_from_file_location(f'sda.{name}', os.path.join(REF, 'sda', f'{name}.py'))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f'sda.{name}'] = mod
        spec.loader.exec_module(mod)
        mods[name] = mod
    return mods['nn'], mods['score']


def _np(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


def _save(name, **arrays):
    flat = {}
    for k, v in arrays.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                flat[f'{k}/{kk}'] = np.asarray(vv)
        else:
            flat[k] = np.asarray(v.detach().cpu().numpy() if torch.is_tensor(v) else v)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **flat)
    print(f'wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB, {len(flat)} arrays)')


def main():
    rnn, rscore = _load_reference()
    SiLU = torch.nn.SiLU

    # ---------------------------------------------------------------- schedule table
    sde0 = rscore.VPSDE(torch.nn.Identity(), shape=())
    tt = torch.linspace(1, 0, 5)
    sub = rscore.SubVPSDE(torch.nn.Identity(), shape=())
    subsub = rscore.SubSubVPSDE(torch.nn.Identity(), shape=())
    lin = rscore.VPSDE(torch.nn.Identity(), shape=(), alpha='lin')
    exp = rscore.VPSDE(torch.nn.Identity(), shape=(), alpha='exp')
    _save('schedule', t=tt, mu_cos=sde0.mu(tt), sigma_cos=sde0.sigma(tt),
          sigma_sub=sub.sigma(tt), sigma_subsub=subsub.sigma(tt),
          mu_lin=lin.mu(tt), sigma_lin=lin.sigma(tt), mu_exp=exp.mu(tt), sigma_exp=exp.sigma(tt))

    # ---------------------------------------------------------------- unfold / fold
    xl = torch.arange(2 * 7 * 3 * 2, dtype=torch.float32).reshape(2, 7, 3, 2)
    for k in (1, 2):
        u = rscore.MCScoreNet.unfold(xl, k)
        f = rscore.MCScoreNet.fold(u, k)
        assert torch.equal(f, xl)
        _save(f'fold_k{k}', x=xl, unfolded=u, refolded=f)

    # ---------------------------------------------------------------- (i) 1-D ScoreUNet, Lorenz-style wrapper
    torch.manual_seed(0)
    net1 = rscore.MCScoreWrapper(rscore.ScoreUNet(3, embedding=8, hidden_channels=(8,), hidden_blocks=(1,),
                                                  activation=SiLU, spatial=1))
    x = torch.randn(2, 16, 3)
    t = torch.tensor(0.37)
    with torch.no_grad():
        out = net1(x, t)
    _save('unet1d_tiny', sd=_np(net1.state_dict()), x=x, t=t, out=out)

    # two-level 1-D net with stride-2 head / upsample tail, zero padding, odd length
    torch.manual_seed(1)
    net1b = rscore.ScoreUNet(3, embedding=8, hidden_channels=(8, 16), hidden_blocks=(1, 2),
                             activation=SiLU, spatial=1)
    xb = torch.randn(3, 3, 20)
    tb = torch.tensor([0.1, 0.5, 0.9])          # per-sample times (training-style call)
    with torch.no_grad():
        outb = net1b(xb, tb)
    _save('unet1d_two_level', sd=_np(net1b.state_dict()), x=xb, t=tb, out=outb)

    # ---------------------------------------------------------------- (ii) 2-D MCScoreNet, LocalScoreUNet, circular
    class LocalScoreUNet(rscore.ScoreUNet):          # same construction as experiments/kolmogorov/utils.py:29-46
        def __init__(self, channels, size=64, **kw):
            super().__init__(channels, 1, **kw)
            domain = 2 * torch.pi / size * (torch.arange(size) + 1 / 2)
            self.register_buffer('forcing', torch.sin(4 * domain).expand(1, size, size).clone())

        def forward(self, x, t, c=None):
            return super().forward(x, t, self.forcing)

    torch.manual_seed(2)
    net2 = rscore.MCScoreNet(2, order=1)
    net2.kernel = LocalScoreUNet(channels=6, size=8, embedding=8, hidden_channels=(4, 8), hidden_blocks=(1, 1),
                                 kernel_size=3, activation=SiLU, spatial=2, padding_mode='circular')
    x2 = torch.randn(2, 5, 2, 8, 8)
    t2 = torch.tensor(0.61)
    taps = {}
    hooks = []
    netw = net2.kernel.network

    def _tap(name):
        def hook(m, i, o):
            taps[name] = o.detach()
        return hook

    hooks.append(netw.heads[0].register_forward_hook(_tap('head0')))
    hooks.append(netw.heads[1].register_forward_hook(_tap('head1')))
    hooks.append(netw.descent[0][0].register_forward_hook(_tap('descent0.0')))
    def _ln_hook(m, i, o):
        taps['ln_in'] = i[0].detach()
        taps['ln_out'] = o.detach()

    def _tap(name):
        def hook(m, i, o):
            taps[name] = o.detach()
        return hook

    hooks.append(netw.descent[0][0].residue[0].register_forward_hook(_ln_hook))
    hooks.append(netw.tails[0].register_forward_hook(_tap('tail1_conv')))
    hooks.append(net2.kernel.embedding.register_forward_hook(_tap('t_emb')))
    with torch.no_grad():
        out2 = net2(x2, t2)
    for h in hooks:
        h.remove()

    # guided score and its gradient on the same net
    def A(x):
        return x[..., ::2, :, ::2, ::2]

    torch.manual_seed(3)
    y_obs = torch.randn(A(x2).shape)
    inner = rscore.VPSDE(net2, shape=())
    gs = rscore.GaussianScore(y_obs, A=A, std=0.5, sde=inner, gamma=1e-2)
    t3 = torch.tensor(0.8)
    # the reference's own d log p / d x (the `s` of score.py:394) is captured from its torch.autograd.grad call
    captured = []
    real_grad = torch.autograd.grad

    def capturing_grad(*a, **kw):
        out = real_grad(*a, **kw)
        captured.append(out[0].detach().clone())
        return out

    torch.autograd.grad = capturing_grad
    try:
        guided = gs(x2, t3)
    finally:
        torch.autograd.grad = real_grad
    assert len(captured) == 1
    grad_logp_ref = captured[0]
    with torch.no_grad():
        plain = net2(x2, t3)
    mu, sigma = inner.mu(t3), inner.sigma(t3)
    grad_logp = (plain - guided) / sigma          # s = d log_p / d x   (score.py:394-396)

    dps = rscore.DPSGaussianScore(y_obs, A=A, sde=inner, zeta=1.0)
    dps_out = dps(x2, t3)

    # one full PC step (predictor + 1 corrector) through the reference's own sample() with injected noise
    sde_g = rscore.VPSDE(gs, shape=(5, 2, 8, 8))
    torch.manual_seed(4)
    steps, corr, tau = 4, 1, 0.5
    state = torch.random.get_rng_state()
    x_init = torch.randn((2,) + (5, 2, 8, 8))
    zs = [torch.randn_like(x_init) for _ in range(steps * corr)]
    torch.random.set_rng_state(state)
    x_final = sde_g.sample((2,), steps=steps, corrections=corr, tau=tau)

    _save('mcscore2d_tiny', sd=_np(net2.state_dict()), x=x2, t=t2, out=out2, taps={k: v for k, v in taps.items()},
          y_obs=y_obs, t_guided=t3, guided=guided, plain=plain, grad_logp=grad_logp, grad_logp_ref=grad_logp_ref,
          dps=dps_out,
          pc_x_init=x_init, pc_noise=torch.stack(zs), pc_x_final=x_final,
          pc_args=np.array([steps, corr, tau]))

    # ---------------------------------------------------------------- (iii) Lorenz local ScoreNet / ResMLP
    torch.manual_seed(5)
    net3 = rscore.MCScoreNet(features=3, order=2, embedding=8, hidden_features=[16] * 2, activation=SiLU)
    x3 = torch.randn(2, 9, 3)
    t4 = torch.tensor(0.25)
    with torch.no_grad():
        out3 = net3(x3, t4)
    _save('scorenet_local_tiny', sd=_np(net3.state_dict()), x=x3, t=t4, out=out3)

    # ---------------------------------------------------------------- unguided sampling, Lorenz global net, 8 steps, 2 corrections
    sde_u = rscore.VPSDE(net1, shape=(16, 3))
    torch.manual_seed(6)
    state = torch.random.get_rng_state()
    xi = torch.randn((3, 16, 3))
    zs = [torch.randn_like(xi) for _ in range(8 * 2)]
    torch.random.set_rng_state(state)
    xf = sde_u.sample((3,), steps=8, corrections=2, tau=0.25)
    _save('sample_unguided_lorenz', x_init=xi, noise=torch.stack(zs), x_final=xf, args=np.array([8, 2, 0.25]))

    # ---------------------------------------------------------------- observation operators of the notebooks
    # KolmogorovFlow.coarsen / .vorticity (sda/mcs.py:340-347, 361-375) are pure-torch static methods, but sda/mcs.py imports
    # jax at module level and cannot be executed here: the two function definitions are extracted from the file's AST and
    # compiled on their own (the reference's text runs; nothing of it is stored).
    import ast
    src = open(os.path.join(REF, 'sda', 'mcs.py')).read()
    tree = ast.parse(src)
    fns = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.ClassDef) and node.name == 'KolmogorovFlow':
            for item in node.body:
                if isinstance(item, ast.FunctionDef) and item.name in ('coarsen', 'vorticity'):
                    item.decorator_list = []
                    mod = ast.Module(body=[item], type_ignores=[])
                    ns = {'torch': torch, 'Tensor': torch.Tensor}
                    exec(compile(ast.fix_missing_locations(mod), 'sda/mcs.py', 'exec'), ns)
                    fns[item.name] = ns[item.name]
    assert set(fns) == {'coarsen', 'vorticity'}
    torch.manual_seed(7)
    xo = torch.randn(2, 3, 2, 16, 24)
    obs = dict(x=xo)
    for r in (2, 4):
        xr = xo.clone().requires_grad_(True)
        out = fns['coarsen'](xr, r)
        cot = torch.randn(out.shape)
        g, = torch.autograd.grad(out, xr, cot)
        obs.update({f'coarsen{r}': out.detach(), f'coarsen{r}_cot': cot, f'coarsen{r}_vjp': g})
    xr = xo.clone().requires_grad_(True)
    out = fns['vorticity'](xr)
    cot = torch.randn(out.shape)
    g, = torch.autograd.grad(out, xr, cot)
    obs.update(vorticity=out.detach(), vorticity_cot=cot, vorticity_vjp=g)
    xr = xo.clone().requires_grad_(True)
    out = fns['vorticity'](fns['coarsen'](xr, 2))           # composition used in figures.ipynb
    cot = torch.randn(out.shape)
    g, = torch.autograd.grad(out, xr, cot)
    obs.update(vort_of_coarsen2=out.detach(), vort_of_coarsen2_cot=cot, vort_of_coarsen2_vjp=g)
    _save('observe_ops', **obs)

    # ---------------------------------------------------------------- key inventory of the real K64 Kolmogorov net
    k64 = rscore.MCScoreNet(2, order=2)
    k64.kernel = LocalScoreUNet(channels=10, embedding=64, hidden_channels=(96, 192, 384), hidden_blocks=(3, 3, 3),
                                kernel_size=3, activation=SiLU, spatial=2, padding_mode='circular')
    keys = {k: np.array(v.shape) for k, v in k64.state_dict().items()}
    nparam = sum(p.numel() for p in k64.parameters())
    _save('k64_keys', shapes=keys, nparam=np.array(nparam))
    print('K64 params:', nparam, 'keys:', len(keys))

    lor = rscore.MCScoreWrapper(rscore.ScoreUNet(channels=3, embedding=32, hidden_channels=(64,), hidden_blocks=(3,),
                                                 activation=SiLU, spatial=1))
    _save('lorenz_global_keys', shapes={k: np.array(v.shape) for k, v in lor.state_dict().items()},
          nparam=np.array(sum(p.numel() for p in lor.parameters())))


if __name__ == '__main__':
    main()
