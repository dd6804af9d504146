#!/usr/bin/env python3 -B
"""Drop-in fixture: the reference's DRIVER FILES, unmodified, on the reference itself and on sda_amd.

Build container only (needs /root/reference); writes `dropin_drivers.npz` (data only).  Two phases, each in its own
interpreter because they bind the name ``sda`` differently:

``--phase ref``   `experiments/lorenz/utils.py` and `experiments/kolmogorov/utils.py` are executed from where they lie on the
                  reference's own `sda/nn.py`, `sda/score.py`, `sda/utils.py` (zuko stand-in exactly as make_golden.py;
                  `sda.mcs` = the jax-free names of sda_amd/mcs.py loaded as a bare file -- the simulators are never called;
                  `h5py`, `ot`, `seaborn` are empty stubs: nothing on this path touches them).  Their own factories build
                  the nets (`make_global_score`, `make_local_score`, `make_score`), a `config.json` + `state.pth` pair is
                  written the way `train.py` does and read back through their own `load_score`, and the posterior-sampling
                  construction of `experiments/lorenz/eval.py:72-84` (Lorenz) / `figures.ipynb#cell9-10` (Kolmogorov) runs
                  for 2 diffusion steps with one correction through the reference's `sample()`.  Inputs, weights, recorded
                  noise and the samples go into the fixture.
``--phase amd``   the same driver files, again unmodified, with ``sda_amd.install_as_sda()`` and the TEST-ONLY CPU shim:
                  same factories, same `load_score`, same construction, noise injected in call order -- compared with the
                  fixture (this is what tests/test_dropin_drivers.py repeats on every CPU run where /root/reference exists).

Usage:  python3 -B tests/golden/make_golden_dropin.py            (ref, then amd as a check)
"""
import argparse
import json
import os
import subprocess
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)

#: reduced widths (the fixture carries the weights); the drivers' defaults otherwise
CONFIGS = {
    'lorenz_global': dict(embedding=8, hidden_channels=[16], hidden_blocks=[2], activation='SiLU'),
    'lorenz_local': dict(window=5, embedding=8, width=32, depth=2, activation='SiLU'),
    'kolmogorov': dict(window=3, embedding=8, hidden_channels=[4, 8], hidden_blocks=[1, 1], kernel_size=3,
                       activation='SiLU'),
}
STEPS, CORR = 2, 1


def jobs(lorenz, kolmogorov, kf):
    """name -> (driver module, factory kwargs for load_score, event shape, batch, A, y-seed, std, gamma, tau).
    Lorenz: experiments/lorenz/eval.py:72-84 with the 'lo' observation (step 8, sigma 0.05, gamma 3e-2, tau 0.25).
    Kolmogorov: figures.ipynb#cell9-10 (every 2nd frame coarsened x8, std 0.1, default gamma, tau 0.5)."""
    step = 8
    return {
        'lorenz_global': (lorenz, dict(local=False), (65, 3), 3, lambda x: x[..., ::step, :1], 0.05, 3e-2, 0.25),
        'lorenz_local': (lorenz, dict(local=True), (65, 3), 3, lambda x: x[..., ::step, :1], 0.05, 3e-2, 0.25),
        'kolmogorov': (kolmogorov, dict(), (4, 2, 64, 64), 1, lambda x: kf.coarsen(x[..., ::2, :, :, :], 8), 0.1, 1e-2, 0.5),
    }


def run_jobs(lorenz, kolmogorov, kf, VPSDE, GaussianScore, states, record):
    """Shared by both phases.  `states`: name -> state_dict to load (None: keep the factory's seeded init and report it).
    `record(name, sde, event, batch, tau)` runs the sampler and returns x."""
    import tempfile
    from pathlib import Path
    out = {}
    for i, (name, (drv, kw, event, batch, A, std, gamma, tau)) in enumerate(jobs(lorenz, kolmogorov, kf).items()):
        torch.manual_seed(100 + i)
        make = {'lorenz_global': getattr(drv, 'make_global_score', None), 'lorenz_local': getattr(drv, 'make_local_score', None),
                'kolmogorov': getattr(drv, 'make_score', None)}[name]
        score = make(**CONFIGS[name])
        if states[name] is not None:
            score.load_state_dict(states[name])
        with tempfile.TemporaryDirectory() as run:               # train.py: save_config(config, runpath); torch.save(state_dict)
            with open(os.path.join(run, 'config.json'), 'x') as f:
                json.dump(CONFIGS[name], f)
            torch.save(score.state_dict(), os.path.join(run, 'state.pth'))
            loaded = drv.load_score(Path(run) / 'state.pth', **kw)
        torch.manual_seed(200 + i)
        y = torch.randn(A(torch.zeros(event)).shape)
        sde = VPSDE(GaussianScore(y=y, A=A, std=std, sde=VPSDE(loaded, shape=()), gamma=gamma), shape=event)
        out[name] = dict(sd=loaded.state_dict(), y=y, **record(name, sde, event, batch, tau))
    return out


# --------------------------------------------------------------------------------------------------------- phase: reference
def phase_ref():
    import importlib.util
    from tests.golden.make_golden import _install_zuko
    from tests import dropin_util as D
    _install_zuko()
    for stub in ('h5py', 'ot', 'seaborn'):
        sys.modules[stub] = types.ModuleType(stub)
    pkg = types.ModuleType('sda')
    pkg.__path__ = [os.path.join(REF, 'sda')]
    sys.modules['sda'] = pkg

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod

    load('sda.mcs', os.path.join(ROOT, 'sda_amd', 'mcs.py'))          # jax-free names only; no simulator is called
    for name in ('nn', 'score', 'utils'):
        load(f'sda.{name}', os.path.join(REF, 'sda', f'{name}.py'))
    rscore = sys.modules['sda.score']
    assert 'sda_amd' not in sys.modules
    with D.scratch_cwd():
        lorenz = D.exec_driver('lorenz', '_ref_lorenz_utils')
        kolmogorov = D.exec_driver('kolmogorov', '_ref_kolmogorov_utils')

        def record(name, sde, event, batch, tau):
            torch.manual_seed(300)
            state = torch.random.get_rng_state()
            x_init = torch.randn((batch,) + event)
            zs = torch.stack([torch.randn_like(x_init) for _ in range(STEPS * CORR)])
            torch.random.set_rng_state(state)
            x = sde.sample((batch,), steps=STEPS, corrections=CORR, tau=tau)
            return dict(x_init=x_init, noise=zs, x_final=x, tau=np.array(tau))

        res = run_jobs(lorenz, kolmogorov, sys.modules['sda.mcs'].KolmogorovFlow, rscore.VPSDE, rscore.GaussianScore,
                       {k: None for k in CONFIGS}, record)
    assert 'sda_amd' not in sys.modules
    flat = {'args': np.array([STEPS, CORR])}
    for name, d in res.items():
        for k, v in d.items():
            if isinstance(v, dict):
                for kk, vv in v.items():
                    flat[f'{name}.sd/{kk}'] = vv.detach().cpu().numpy()
            else:
                flat[f'{name}.{k}'] = v.detach().cpu().numpy() if torch.is_tensor(v) else v
    path = os.path.join(HERE, 'dropin_drivers.npz')
    np.savez_compressed(path, **flat)
    print(f'wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB, {len(flat)} arrays)')


# ------------------------------------------------------------------------------------------------------------ phase: sda_amd
def load_fixture():
    data = np.load(os.path.join(HERE, 'dropin_drivers.npz'))
    out = {}
    for k in data.files:
        if k == 'args':
            continue
        name, rest = k.split('.', 1)
        d = out.setdefault(name, {'sd': {}})
        v = torch.from_numpy(np.asarray(data[k]))
        if rest.startswith('sd/'):
            d['sd'][rest[3:]] = v
        else:
            d[rest] = v
    return out, [int(v) for v in data['args']]


def phase_amd(tol=1e-4):
    """Returns name -> max |x - x_ref| / max |x_ref|; raises when a job leaves the tolerance."""
    import pytest
    from tests import cpu_shim, dropin_util as D
    from tests.util import assert_close
    fx, (steps, corr) = load_fixture()
    mp = pytest.MonkeyPatch()
    errs = {}
    try:
        cpu_shim.install(mp)
        with D.driver('lorenz') as lorenz, D.driver('kolmogorov') as kolmogorov:
            import sda_amd
            from sda_amd.score import GaussianScore, VPSDE

            def record(name, sde, event, batch, tau):
                assert abs(tau - float(fx[name]['tau'])) < 1e-12
                zs = fx[name]['noise']
                sde.initial_noise = fx[name]['x_init']
                sde.noise_source = lambda i, j: zs[i * corr + j]
                return dict(x_final=sde.sample((batch,), steps=steps, corrections=corr, tau=tau))

            res = run_jobs(lorenz, kolmogorov, sda_amd.mcs.KolmogorovFlow, VPSDE, GaussianScore,
                           {k: fx[k]['sd'] for k in CONFIGS}, record)
        for name, d in res.items():
            assert list(d['sd']) == list(fx[name]['sd']), f'{name}: state_dict keys differ from the reference'
            assert torch.equal(d['y'], fx[name]['y'])
            ref = fx[name]['x_final']
            errs[name] = ((d['x_final'] - ref).abs().max() / ref.abs().max()).item()
            assert_close(d['x_final'], ref, tol, what=name)
    finally:
        mp.undo()
    return errs


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--phase', choices=('ref', 'amd', 'both'), default='both')
    a = ap.parse_args()
    torch.set_num_threads(8)
    if a.phase == 'both':
        for ph in ('ref', 'amd'):
            subprocess.run([sys.executable, '-B', os.path.abspath(__file__), '--phase', ph], check=True, cwd=ROOT)
    elif a.phase == 'ref':
        phase_ref()
    else:
        for k, v in phase_amd().items():
            print(f'{k}: sda_amd on the reference drivers vs the reference itself, rel err {v:.2e}')
