"""GPU: the drop-in fixture (tests/golden/dropin_drivers.npz -- the reference's own driver files run on the reference itself
in the build container) through the HIP path, and the routes a reference-style `LocalScoreUNet` takes."""
import pytest
import torch
import torch.nn as nn

from tests.golden import make_golden_dropin as G
from tests.test_dropin_drivers import run_fixture_job
from tests.util import assert_close, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    from sda_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


@pytest.mark.parametrize('name', ['lorenz_global', 'lorenz_local', 'kolmogorov'])
def test_dropin_fixture_on_device(dev, name):
    """`make_*_score` + `load_state_dict` + the eval.py:72-84 / figures.ipynb#cell9-10 construction + two PC steps with one
    correction, against what the reference produced with the same driver files, weights and noise (rtol 1e-4, fp32)."""
    fx, (steps, corr) = G.load_fixture()
    x = run_fixture_job(name, fx, steps, corr, device=dev)
    assert x.is_cuda
    assert_close(x.cpu(), fx[name]['x_final'], 1e-4, what=name)


def test_reference_style_subclass_takes_the_fused_route(dev, monkeypatch):
    """A `LocalScoreUNet` written as experiments/kolmogorov/utils.py:29-46 writes it (forward override -> self.forcing) must
    give what the stock class with a `_context` hook gives, bit for bit, value and input-VJP -- same kernels, same order --
    and agree with the generic route (materialised unfold, context concat, autograd) to fp32 round-off."""
    from sda_amd import score as S
    kw = dict(embedding=16, hidden_channels=(8, 16), hidden_blocks=(1, 2), kernel_size=3, activation=nn.SiLU, spatial=2,
              padding_mode='circular')
    size = 16

    def forcing():
        domain = 2 * torch.pi / size * (torch.arange(size) + 1 / 2)
        return torch.sin(4 * domain).expand(1, size, size).clone()

    class RefStyle(S.ScoreUNet):
        def __init__(self, channels, **kwargs):
            super().__init__(channels, 1, **kwargs)
            self.register_buffer('forcing', forcing())

        def forward(self, x, t, c=None):
            return super().forward(x, t, self.forcing)

    class Hooked(S.ScoreUNet):                          # the stock forward; context through the package's hook
        def __init__(self, channels, **kwargs):
            super().__init__(channels, 1, **kwargs)
            self.register_buffer('forcing', forcing())

        def _context(self, c):
            return self.forcing

    class Generic(RefStyle):                            # post-processes: not context-only
        def forward(self, x, t, c=None):
            return super().forward(x, t, c) * 1.0

    routes = []
    real = S._MCScoreFunction.apply
    monkeypatch.setattr(S._MCScoreFunction, 'apply', staticmethod(lambda *a: (routes.append('fused'), real(*a))[1]))
    torch.manual_seed(0)
    nets = {}
    for cls in (RefStyle, Hooked, Generic):
        net = S.MCScoreNet(2, order=2)
        net.kernel = cls(10, **kw)
        if nets:
            net.load_state_dict(nets['RefStyle'].state_dict())
        nets[cls.__name__] = net.to(dev)
    x = torch.randn(2, 7, 2, size, size, device=dev)
    t = torch.tensor(0.43, device=dev)
    g = torch.randn_like(x)
    res = {}
    for k, net in nets.items():
        n0 = len(routes)
        xs = x.clone().requires_grad_(True)
        out = net(xs, t)
        v, = torch.autograd.grad(out, xs, g)
        res[k] = (out.detach(), v, len(routes) - n0)
    assert res['RefStyle'][2] == 1 and res['Hooked'][2] == 1 and res['Generic'][2] == 0
    assert torch.equal(res['RefStyle'][0], res['Hooked'][0]) and torch.equal(res['RefStyle'][1], res['Hooked'][1])
    assert_close(res['RefStyle'][0], res['Generic'][0], 1e-6, atol=1e-6, what='value, fused vs generic')
    assert_close(res['RefStyle'][1], res['Generic'][1], 1e-5, what='VJP, fused vs generic')
