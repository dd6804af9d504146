"""CPU: the drop-in boundary for the reference's DRIVER FILES (BASELINE north_star: "drops into
experiments/{lorenz,kolmogorov} unchanged").

* where /root/reference exists (the build container), `experiments/lorenz/utils.py` and `experiments/kolmogorov/utils.py`
  are executed UNMODIFIED on sda_amd (`sda_amd.install_as_sda()`, TEST-ONLY CPU shim for the kernels): their factories,
  their `load_score`, the `eval.py:72-84` / `figures.ipynb#cell9-10` construction and `sample()` -- against
  tests/golden/dropin_drivers.npz, which tests/golden/make_golden_dropin.py produced by running the SAME driver files on
  the reference's own `sda/{nn,score,utils}.py`;
* everywhere, the same fixture through this package's mirrored factories (`sda_amd.experiments`), the recognition of
  context-only `forward` overrides (the reference's `LocalScoreUNet`), the `sda.mcs` passthrough and the snippet of
  INTEGRATION.md section 1.
"""
import os
import re
import subprocess
import sys

import pytest
import torch
import torch.nn as nn

from tests import cpu_shim, dropin_util
from tests.golden import make_golden_dropin as G
from tests.util import assert_close, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needs_reference = pytest.mark.skipif(not dropin_util.have_reference(), reason='/root/reference is not present (GPU box)')


@pytest.fixture
def shim(monkeypatch):
    cpu_shim.install(monkeypatch)


@needs_reference
def test_reference_driver_files_unmodified_on_sda_amd(monkeypatch):
    from sda_amd import score as S
    seen = []
    real = S._context_only_override

    def spy(kernel, shape, t, c):
        out = real(kernel, shape, t, c)
        seen.append((type(kernel).__module__, out[0], out[1] is getattr(kernel, 'forcing', None)))
        return out

    monkeypatch.setattr(S, '_context_only_override', spy)
    errs = G.phase_amd(tol=1e-4)
    assert set(errs) == {'lorenz_global', 'lorenz_local', 'kolmogorov'}
    # the Kolmogorov kernel is the class DEFINED IN THE REFERENCE'S FILE (forward override handing on self.forcing): every
    # evaluation recognised it as context-only and took the fused window path with the forcing plane as context
    ref_defined = [s for s in seen if s[0] == '_reference_kolmogorov_utils']
    assert ref_defined and all(ok and is_forcing for _, ok, is_forcing in ref_defined)
    assert 'sda' not in sys.modules or sys.modules['sda'].__name__ != 'sda_amd'      # aliasing undone


def _mirrored(name, sd):
    from sda_amd.experiments import kolmogorov, lorenz
    cfg = G.CONFIGS[name]
    make = {'lorenz_global': lorenz.make_global_score, 'lorenz_local': lorenz.make_local_score,
            'kolmogorov': kolmogorov.make_score}[name]
    score = make(**cfg)
    score.load_state_dict(sd)
    return score


def run_fixture_job(name, fx, steps, corr, device=None, use_graph=False):
    """The construction make_golden_dropin.jobs() describes, on this package's mirrored factories."""
    from sda_amd import mcs
    from sda_amd.score import GaussianScore, VPSDE
    _, kw, event, batch, A, std, gamma, tau = G.jobs(None, None, mcs.KolmogorovFlow)[name]
    score = _mirrored(name, fx[name]['sd'])
    sde = VPSDE(GaussianScore(y=fx[name]['y'], A=A, std=std, sde=VPSDE(score, shape=()), gamma=gamma), shape=event)
    if device is not None:
        sde = sde.to(device)
    zs = fx[name]['noise']
    sde.initial_noise = fx[name]['x_init']
    sde.noise_source = lambda i, j: zs[i * corr + j]
    return sde.sample((batch,), steps=steps, corrections=corr, tau=float(fx[name]['tau']))


@pytest.mark.parametrize('name', ['lorenz_global', 'lorenz_local', 'kolmogorov'])
def test_fixture_through_mirrored_factories(shim, name):
    fx, (steps, corr) = G.load_fixture()
    x = run_fixture_job(name, fx, steps, corr)
    assert_close(x, fx[name]['x_final'], 1e-4, what=name)


def test_context_only_override_recognition(shim):
    from sda_amd import score as S
    g, grp = load_golden('mcscore2d_tiny')
    kw = dict(embedding=8, hidden_channels=(4, 8), hidden_blocks=(1, 1), kernel_size=3, activation=nn.SiLU, spatial=2,
              padding_mode='circular')

    def forcing(size):
        domain = 2 * torch.pi / size * (torch.arange(size) + 1 / 2)
        return torch.sin(4 * domain).expand(1, size, size).clone()

    class RefStyle(S.ScoreUNet):                       # experiments/kolmogorov/utils.py:29-46, as the reference writes it
        def __init__(self, channels, size=64, **kwargs):
            super().__init__(channels, 1, **kwargs)
            self.register_buffer('forcing', forcing(size))

        def forward(self, x, t, c=None):
            return super().forward(x, t, self.forcing)

    class Rescaled(RefStyle):                          # touches x: NOT context-only
        def forward(self, x, t, c=None):
            return super().forward(x * 1.0, t, c)

    class PostProcessed(RefStyle):                     # touches the result
        def forward(self, x, t, c=None):
            return super().forward(x, t, c) + 0.0

    class Twice(RefStyle):                             # two network evaluations
        def forward(self, x, t, c=None):
            a = S.ScoreUNet.forward(self, x, t, self.forcing)
            return (a + S.ScoreUNet.forward(self, x, t, self.forcing)) / 2

    class Picky(RefStyle):                             # cannot digest a storage-less placeholder
        def forward(self, x, t, c=None):
            assert x.device.type != 'meta'
            return super().forward(x, t, self.forcing)

    class InPlaceOut(RefStyle):                        # in-place on the result: identity is kept, the version counter is not
        def forward(self, x, t, c=None):
            out = super().forward(x, t, c)
            out.mul_(1.0)
            return out

    class InPlaceIn(RefStyle):                         # in-place on the input before the call
        def forward(self, x, t, c=None):
            x.mul_(1.0)
            return super().forward(x, t, c)

    outs = {}
    for cls, expect in ((RefStyle, True), (Rescaled, False), (PostProcessed, False), (Twice, False), (Picky, False),
                        (InPlaceOut, False), (InPlaceIn, False)):
        net = S.MCScoreNet(2, order=1)
        net.kernel = cls(6, size=8, **kw)
        net.load_state_dict(grp['sd'])
        shape = (2, 3, 6, 8, 8)
        ok, ctx = S._context_only_override(net.kernel, shape, g['t'], None)
        assert ok is expect, cls.__name__
        assert (ctx is net.kernel.forcing) if expect else ctx is None
        assert S._probe.calls is None and S._probe.out is None                 # disarmed again
        with torch.no_grad():
            outs[cls.__name__] = net(g['x'], g['t'])
    for k, v in outs.items():                          # fused and generic routes agree, and with the reference's output
        assert_close(v, g['out'], 2e-5, what=k)
    # a hooked kernel keeps its hooks: generic path
    net = S.MCScoreNet(2, order=1)
    net.kernel = RefStyle(6, size=8, **kw)
    net.load_state_dict(grp['sd'])
    seen = []
    h = net.kernel.register_forward_hook(lambda m, i, o: seen.append(tuple(i[0].shape)))
    with torch.no_grad():
        net(g['x'], g['t'])
    h.remove()
    assert seen == [(2, 3, 6, 8, 8)]


def test_mcs_fallback_names_and_helpers():
    from sda_amd import mcs
    g, _ = load_golden('observe_ops')                  # produced by the reference's KolmogorovFlow.coarsen / .vorticity
    for name in ('MarkovChain', 'NoisyLorenz63', 'Lorenz96', 'KolmogorovFlow', 'Normal', 'np', 'torch', 'Tensor', 'Size',
                 'Callable', 'Sequence'):
        assert hasattr(mcs, name), name
    if mcs.SOURCE is None:
        with pytest.raises(ImportError, match='simulator of the reference package'):
            mcs.NoisyLorenz63(dt=0.025)
    kf = mcs.KolmogorovFlow
    for r in (2, 4):
        x = g['x'].clone().requires_grad_(True)
        out = kf.coarsen(x, r)
        assert torch.allclose(out, g[f'coarsen{r}'], rtol=0, atol=1e-6)
        v, = torch.autograd.grad(out, x, g[f'coarsen{r}_cot'])
        assert torch.allclose(v, g[f'coarsen{r}_vjp'], rtol=0, atol=1e-6)
    x = g['x'].clone().requires_grad_(True)
    out = kf.vorticity(x)
    assert torch.allclose(out, g['vorticity'], rtol=0, atol=1e-6)
    v, = torch.autograd.grad(out, x, g['vorticity_cot'])
    assert torch.allclose(v, g['vorticity_vjp'], rtol=0, atol=1e-6)
    up = kf.upsample(g['x'][0, 0], 2)
    assert up.shape == (2, 32, 48)
    assert torch.allclose(kf.coarsen(kf.upsample(g['x'], 2, mode='nearest'), 2), g['x'], atol=1e-6)


def test_mcs_passthrough_of_a_user_file(tmp_path):
    """`$SDA_MCS_FILE` (or an `sda` package on sys.path) is re-exported when it imports; a failing import (jax absent) degrades
    to the placeholders with the reason recorded."""
    good = tmp_path / 'mcs_good.py'
    good.write_text('import torch\nclass NoisyLorenz63:\n    def __init__(self, dt): self.dt = dt\nMARK = 7\n')
    bad = tmp_path / 'mcs_bad.py'
    bad.write_text('import a_module_that_does_not_exist_anywhere\n')
    code = ('import sys; sys.path.insert(0, %r); import sda_amd.mcs as m; '
            'print(m.SOURCE is not None, getattr(m, "MARK", None), m.NoisyLorenz63.__module__, bool(m.UNAVAILABLE))' % ROOT)
    for path, expect in ((good, 'True 7 sda_amd._user_mcs False'), (bad, 'False None sda_amd.mcs True')):
        env = dict(os.environ, SDA_MCS_FILE=str(path))
        out = subprocess.run([sys.executable, '-B', '-c', code], env=env, capture_output=True, text=True, check=True)
        assert out.stdout.strip() == expect, out.stdout + out.stderr


def test_install_as_sda_registers_single_copies():
    code = ('import sys; sys.path.insert(0, %r); import sda_amd; sda_amd.install_as_sda(); '
            'import sda, sda.score, sda.nn, sda.utils, sda.mcs, sda.observe; from sda.score import VPSDE; '
            'import sda_amd.score; print(sda is sda_amd, sda.score is sda_amd.score, VPSDE is sda_amd.score.VPSDE, '
            'sda.mcs is sda_amd.mcs, sda.observe is sda_amd.observe)' % ROOT)
    out = subprocess.run([sys.executable, '-B', '-c', code], capture_output=True, text=True, check=True)
    assert out.stdout.strip() == 'True True True True True', out.stdout + out.stderr


def _integration_snippets():
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    sec = text.split('## 1.', 1)[1].split('\n## 2.', 1)[0]
    return re.findall(r'```python\n(.*?)```', sec, flags=re.S)


@needs_reference
@pytest.mark.parametrize('name', ['lorenz', 'kolmogorov'])
def test_integration_snippet_runs_verbatim(tmp_path, name):
    """INTEGRATION.md section 1's first python block, character for character, from the root of a reference checkout (a
    scratch directory whose `experiments` is a symlink to the reference's), for both driver files.  Only the plotting
    dependency this image lacks (seaborn) is stubbed, before the snippet starts."""
    snippet = _integration_snippets()[0]
    assert "'experiments/lorenz'" in snippet
    if name == 'kolmogorov':
        snippet = snippet.replace("'experiments/lorenz'", "'experiments/kolmogorov'") \
                         .replace('make_global_score()', 'make_score(window=5)')
    os.symlink('/root/reference/experiments', tmp_path / 'experiments')
    pre = ('import sys, types\nsys.path.insert(0, %r)\n'
           'try:\n    import seaborn\nexcept ImportError:\n    sys.modules["seaborn"] = types.ModuleType("seaborn")\n' % ROOT)
    post = '\nprint("DROPIN", type(score).__module__, sum(p.numel() for p in score.parameters()))\n'
    out = subprocess.run([sys.executable, '-B', '-c', pre + snippet + post], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('DROPIN')][0].split()
    assert line[1] == 'sda_amd.score'
    assert int(line[2]) == 178787 if name == 'lorenz' else int(line[2]) > 1_000_000      # (lorenz/utils.py:26-42 defaults)
