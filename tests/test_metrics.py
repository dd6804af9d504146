"""Evaluation metrics: oracle vs the reference's own outputs, HIP path vs oracle.

The reference's `emd` needs POT, which is absent: emd has no reference-generated fixture ("parity unpinned"); it is
anchored on hand-checkable cases (1-D optimal transport = sorted matching) and on scipy's assignment / LP solvers."""
import numpy as np
import pytest
import torch

from oracle import sda_oracle as O
from tests.util import assert_close, load_golden


@pytest.fixture(scope='module')
def dev():
    from sda_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


def _bpf_callables(g):
    rot = g['rot']

    def transition(x):
        return x @ rot.T + 0.1 * torch.randn_like(x)

    def likelihood(yi, x):
        return torch.softmax(-((x[:, :1] - yi) ** 2).sum(-1) / 0.5, 0)
    return transition, likelihood


# ------------------------------------------------------------------------------------------------ CPU: oracle + host code
def test_oracle_mmd_golden():
    g, _ = load_golden('metrics_mmd')
    assert_close(O.mmd(g['x'], g['y']), g['mmd_xy'], 1e-5)
    assert_close(O.mmd(g['x'], g['x'][:24]), g['mmd_xx'], 1e-5)
    # the exact (float64, difference-based) value sits within the reference's own fp32 noise
    assert abs(float(O.mmd(g['x'], g['y'], exact=True)) - float(g['mmd_xy'])) < 2e-4 * 7


def test_oracle_and_host_bpf_golden():
    from sda_amd.metrics import bpf
    g, _ = load_golden('metrics_bpf')
    tr, lk = _bpf_callables(g)
    for fn in (O.bpf, bpf):                      # the product's bpf is device-agnostic torch code: exact on CPU too
        torch.manual_seed(int(g['seed']))
        out = fn(g['x0'], g['y'], tr, lk, step=int(g['step']))
        assert out.shape == g['out'].shape
        assert torch.equal(out, g['out'])


def test_oracle_emd_known_answers():
    # 1-D: optimal transport between equally weighted point sets is the sorted matching
    torch.manual_seed(3)
    x, y = torch.randn(37, 1), torch.randn(37, 1) * 2 + 1
    want = (x.flatten().sort().values - y.flatten().sort().values).abs().mean()
    assert_close(O.emd(x, y), want, 1e-6)
    # translation of a point cloud: every point moves by the shift
    z = torch.randn(20, 4, 3)
    assert_close(O.emd(z, z + 0.5), torch.tensor(0.5 * (12 ** 0.5)), 1e-5)
    # LP path (unequal counts) agrees with the assignment path when one set is the other with every point doubled
    a, b = torch.randn(6, 2), torch.randn(6, 2)
    assert_close(O.emd(a.repeat(2, 1), b), O.emd(a, b), 1e-6)


def test_host_assignment_solver_matches_scipy():
    from scipy.optimize import linear_sum_assignment
    from sda_amd import ops
    rng = np.random.default_rng(5)
    for n in (1, 2, 7, 64, 193):
        c = torch.from_numpy(rng.random((n, n)).astype(np.float32) * 10)
        total, cols = ops.assignment_cost(c)
        r, cidx = linear_sum_assignment(c.double().numpy())
        assert abs(total - c.double().numpy()[r, cidx].sum()) < 1e-9 * max(1.0, total)
        assert sorted(cols.tolist()) == list(range(n))                    # a permutation
        assert abs(c.double()[torch.arange(n), cols.long()].sum().item() - total) < 1e-9 * max(1.0, total)
    # ties / degenerate costs
    total, cols = ops.assignment_cost(torch.ones(5, 5))
    assert total == 5.0 and sorted(cols.tolist()) == list(range(5))


def test_host_transport_solver_matches_linprog_and_replication():
    """emd for M != N: the host min-cost-flow solve against scipy's LP solver (HiGHS) on the transport LP itself, against
    the assignment solver on the lcm-replicated problem (uniform marginals: exact), and the 1-D closed form."""
    from scipy.optimize import linprog
    from sda_amd import ops
    rng = np.random.default_rng(11)

    def lp(c):
        m, n = c.shape
        a_eq = np.zeros((m + n, m * n))
        for i in range(m):
            a_eq[i, i * n:(i + 1) * n] = 1
        for j in range(n):
            a_eq[m + j, j::n] = 1
        b_eq = np.concatenate([np.full(m, 1 / m), np.full(n, 1 / n)])
        return linprog(c.reshape(-1), A_eq=a_eq, b_eq=b_eq, bounds=(0, None), method='highs').fun

    for m, n in ((1, 1), (1, 7), (5, 1), (2, 3), (7, 5), (12, 18), (31, 64), (40, 33)):
        c = rng.random((m, n)).astype(np.float32) * 4
        got = ops.transport_cost(torch.from_numpy(c))
        assert abs(got - lp(c.astype(np.float64))) < 1e-9 * max(1.0, got), (m, n)
        # every row n/g times, every column m/g times (g = gcd): an assignment problem with the same optimum
        g = int(np.gcd(m, n))
        rep = np.repeat(np.repeat(c, n // g, axis=0), m // g, axis=1)
        tot, _ = ops.assignment_cost(torch.from_numpy(np.ascontiguousarray(rep)))
        assert abs(got - tot / rep.shape[0]) < 1e-9 * max(1.0, got), (m, n)
    # square input: the same value as the assignment path
    c = rng.random((23, 23)).astype(np.float32)
    assert abs(ops.transport_cost(torch.from_numpy(c)) - ops.assignment_cost(torch.from_numpy(c))[0] / 23) < 1e-12
    # 1-D: W1 = integral |F^-1 - G^-1| of the two empirical quantile functions
    x, y = np.sort(rng.normal(size=9)), np.sort(rng.normal(size=6) + 1)
    c = np.abs(x[:, None] - y[None, :]).astype(np.float32)
    grid = (np.arange(18) + 0.5) / 18
    w1 = np.abs(x[np.minimum((grid * 9).astype(int), 8)] - y[np.minimum((grid * 6).astype(int), 5)]).mean()
    assert abs(ops.transport_cost(torch.from_numpy(c)) - w1) < 1e-6
    # ties, and rejected input
    assert ops.transport_cost(torch.ones(4, 6)) == pytest.approx(1.0, abs=1e-12)
    from sda_amd._lib import SdaHipError
    with pytest.raises(SdaHipError):
        ops.transport_cost(torch.tensor([[1.0, float('nan')]]))


def test_metrics_reject_cpu_tensors():
    from sda_amd._lib import SdaHipError
    from sda_amd.metrics import emd, mmd
    with pytest.raises(SdaHipError):
        mmd(torch.randn(4, 3), torch.randn(5, 3))
    with pytest.raises(SdaHipError):
        emd(torch.randn(4, 3), torch.randn(4, 3))


# ------------------------------------------------------------------------------------------------ GPU: HIP path vs oracle
@pytest.mark.gpu
def test_pairwise_dist(dev):
    from sda_amd import ops
    torch.manual_seed(0)
    for m, n, d in ((1, 1, 1), (5, 70, 3), (64, 64, 32), (130, 97, 195), (33, 200, 1000)):
        x, y = torch.randn(m, d), torch.randn(n, d) * 1.5 + 0.3
        ref = torch.cdist(x.double(), y.double())
        assert_close(ops.pairwise_dist(x.to(dev), y.to(dev), squared=False).cpu(), ref.float(), 1e-5)
        assert_close(ops.pairwise_dist(x.to(dev), y.to(dev), squared=True).cpu(), ref.square().float(), 1e-5)


@pytest.mark.gpu
def test_mmd_golden_and_oracle(dev):
    from sda_amd.metrics import mmd
    g, _ = load_golden('metrics_mmd')
    got = mmd(g['x'].to(dev), g['y'].to(dev)).cpu()
    # against the exact value (rtol 1e-4 of the kernel-mean scale, 7 bandwidths) and the reference's own fp32 output
    assert abs(float(got) - float(O.mmd(g['x'], g['y'], exact=True))) < 1e-5
    assert abs(float(got) - float(g['mmd_xy'])) < 2e-4 * 7
    torch.manual_seed(8)
    x, y = torch.randn(300, 65, 3), torch.randn(260, 65, 3) * 1.1
    assert abs(float(mmd(x.to(dev), y.to(dev))) - float(O.mmd(x, y, exact=True))) < 1e-5
    assert float(mmd(x.to(dev), x.to(dev))) == pytest.approx(0.0, abs=1e-6)


@pytest.mark.gpu
def test_emd_vs_oracle(dev):
    from sda_amd.metrics import emd
    torch.manual_seed(9)
    for n, shape in ((37, (1,)), (128, (65, 3)), (512, (16, 3))):
        x, y = torch.randn(n, *shape), torch.randn(n, *shape) * 1.3 + 0.2
        assert_close(emd(x.to(dev), y.to(dev)).cpu(), O.emd(x, y), 1e-5)
    # unequal sample counts: the transport LP (oracle: scipy linprog)
    for (m, n), shape in (((9, 6), (4,)), ((40, 64), (5, 3)), ((96, 37), (16, 2))):
        x, y = torch.randn(m, *shape), torch.randn(n, *shape) * 0.8 - 0.4
        assert_close(emd(x.to(dev), y.to(dev)).cpu(), O.emd(x, y), 1e-5)
    x = torch.randn(64, 1)
    y = torch.randn(64, 1) + 3
    want = (x.flatten().sort().values - y.flatten().sort().values).abs().mean()
    assert_close(emd(x.to(dev), y.to(dev)).cpu(), want, 1e-5)


@pytest.mark.gpu
def test_bpf_on_device(dev):
    """Device run of the filter: shapes, finiteness, and ancestry (every history is a prefix-consistent trajectory)."""
    from sda_amd.metrics import bpf
    g, _ = load_golden('metrics_bpf')
    rot = g['rot'].to(dev)
    torch.manual_seed(1)
    out = bpf(g['x0'].to(dev), g['y'].to(dev), lambda x: x @ rot.T,                 # deterministic transition
              lambda yi, x: torch.softmax(-((x[:, :1] - yi) ** 2).sum(-1) / 0.5, 0), step=2)
    assert out.shape == (64, 11, 2) and torch.isfinite(out).all()
    assert_close(out[:, 1:].cpu(), (out[:, :-1] @ rot.T).cpu(), 1e-5)               # histories moved with their ancestors
