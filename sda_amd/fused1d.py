r"""Two-launch guided score evaluation and six-launch predictor-corrector step for the 1-D (Lorenz) score networks.

The reference's canonical sampling job (experiments/lorenz/eval.py:72-84) is ``VPSDE(GaussianScore(y, A = x[..., ::step, :1], std,
sde = VPSDE(score, shape=()), gamma)).sample(...)`` with ``score = MCScoreWrapper(ScoreUNet(...))`` (lorenz/utils.py:26-42).  One
guided evaluation (sda/score.py:375-396) has exactly one global dependency -- every position of eps feeds the likelihood whose
cotangent feeds the VJP at every position -- so two launches are its minimum:

    sda_step1d_prologue          schedule scalars, time embedding, all modulation vectors (both times of a PC step)
    sda_net1d_fwd_fused          eps (+ saved activations) and the likelihood cotangent ghat = A^T((y - A x_hat) / var)
    sda_net1d_bwd_fused          J_eps^T ghat, the guided score, and by mode: the predictor update in place / the per-tile sums of
                                 squares of the Langevin step size
    sda_pc_correct(_keyed)       the corrector update (row-keyed noise generated in the kernel)

instead of ~14 launches per evaluation + 7 per step.  Every kernel replays the arithmetic of the unfused ones operation for operation
(tests compare the two paths at 1e-6).  Anything outside this shape -- other observation operators, per-sample times, context, a
network the whole-net kernel does not take, detached guidance -- returns ``None`` from :func:`plan` and the general path runs.
"""
import ctypes
import math
import os
from typing import Optional

import torch
from torch import Tensor

from . import _lib, ops
from .engine import Source

ENABLED = os.environ.get('SDA_FUSED1D', '1') != '0'
COEF_LEN = 16                # mu0 sigma0 mu1 sigma1 r c1 sigma_next t0 t1 (+ 7 slots: phase stamps of a -DSDA_S1_TRACE tooling build)


def stock_schedule(sde):
    """(alpha_kind, eta, k, sigma_kind) when ``sde`` computes mu / sigma with the reference's stock formulas (sda/score.py:195-210,
    279-302) -- the condition of VPSDE.mu_sigma's one-launch path --, else None."""
    from .score import _ALPHA_KINDS, _SIGMA_KINDS, VPSDE
    cls = type(sde)
    if not isinstance(sde, VPSDE):
        return None
    kind = _SIGMA_KINDS.get(cls.sigma)
    stock_alpha = sde.alpha is sde._alpha_stock and not any('alpha' in k.__dict__ for k in cls.__mro__)
    if kind is None or cls.mu is not VPSDE.mu or not stock_alpha or sde.alpha_kind not in _ALPHA_KINDS:
        return None
    ak = _ALPHA_KINDS[sde.alpha_kind]
    return ak, float(sde.eta), (0.0, math.acos(math.sqrt(sde.eta)), math.log(sde.eta))[ak], kind


def _slice3(sl: slice, n: int):
    a, b, c = sl.indices(n)
    if c <= 0 or b <= a:
        return None
    return a, c, min(b, n)


class Fused1D:
    """Buffers and descriptors of the fused evaluation for one (GaussianScore, batch shape)."""

    def __init__(self, gs, inner, affine, engine, plan1d, x: Tensor, pos, chan, y_sn: int, sched):
        self.gs, self.inner, self.affine, self.engine, self.plan1d = gs, inner, affine, engine, plan1d
        self.pos, self.chan, self.y_sn, self.sched = pos, chan, y_sn, sched
        B, L, C = x.shape
        self.shape = (B, L, C)
        dev = x.device
        self.dev = dev
        lev, blocks = plan1d['lev'], plan1d['blocks']
        nb = len(blocks)
        self.eps = torch.empty(B, L, C, device=dev, dtype=torch.float32)
        self.ghat = torch.empty_like(self.eps)
        self.out = torch.empty_like(self.eps)
        self.a_s = torch.empty(nb, B, lev.C, L, device=dev, dtype=torch.float32)
        self.z_s = torch.empty_like(self.a_s)
        self.m_s = torch.empty(nb, B, L, device=dev, dtype=torch.float32)
        self.r_s = torch.empty_like(self.m_s)
        self.coef = torch.zeros(COEF_LEN, device=dev, dtype=torch.float32)
        self.step_i = torch.zeros(1, device=dev, dtype=torch.int64)
        self.mod = torch.empty(2, engine.mod_total, device=dev, dtype=torch.float32)
        d, _ = self._desc(False, 0)
        self.ptiles = _lib.load().sda_net1d_tiles(ctypes.byref(d))
        if self.ptiles <= 0:
            raise _lib.SdaHipError('fused1d: the whole-net kernel declined a shape its planner accepted')
        self.partial = torch.empty(B, self.ptiles, device=dev, dtype=torch.float32)

    # ------------------------------------------------------------------ descriptors
    def _desc(self, backward: bool, k: int):
        B, L, C = self.shape
        d, keep = self.engine._net1d_desc(self.plan1d, B, L, self.mod[k:k + 1], 0, False, backward, cin_keep=C if backward else 0)
        lev = self.plan1d['lev']
        d.x_sn = d.out_sn = L * C
        d.x_sc = d.out_sc = 1
        d.x_sx = d.out_sx = C
        d.a_save, d.z_save, d.save_stride = self.a_s.data_ptr(), self.z_s.data_ptr(), B * lev.C * L
        d.mean_save, d.rstd_save, d.stat_stride = self.m_s.data_ptr(), self.r_s.data_ptr(), B * L
        return d, keep

    def _fuse(self, k: int):
        f = _lib.Net1dFuse()
        f.cx0, f.cx1, f.cn = self.affine
        f.coef = self.coef.data_ptr() + 8 * k
        return f

    # ------------------------------------------------------------------ launches
    def _prologue(self, table: Optional[Tensor], istep: Optional[Tensor], t: Optional[Tensor], nt: int):
        emb = self.inner.score.embedding
        w0, b0, w2, b2 = emb[0].weight.detach(), emb[0].bias.detach(), emb[2].weight.detach(), emb[2].bias.detach()
        wp, bp = self.engine.projection()
        ak, eta, kk, sk = self.sched
        ops._dev(w0, b0, w2, b2, wp, bp, emb.freqs, t)
        _lib.check(_lib.load().sda_step1d_prologue(
            ops._ptr(table), 0 if table is None else table.shape[1], ops._ptr(istep), ops._ptr(t), nt, ak, eta, kk, sk,
            emb.freqs.data_ptr(), emb.freqs.numel(), w0.data_ptr(), b0.data_ptr(), w0.shape[0], w2.data_ptr(), b2.data_ptr(), w2.shape[0],
            wp.data_ptr(), bp.data_ptr(), wp.shape[0], self.coef.data_ptr(), self.step_i.data_ptr(), self.mod.data_ptr(), ops._stream()),
            'sda_step1d_prologue')

    def prologue_step(self, table: Tensor, istep: Tensor):
        """This step's scalars and both evaluations' modulation vectors from the device schedule table; advances ``istep``."""
        self._prologue(table, istep, None, 2)

    def prologue_eval(self, t: Tensor):
        self._prologue(None, None, t.reshape(1), 1)

    def forward(self, x: Tensor, k: int):
        d, keep = self._desc(False, k)
        d.x, d.out = x.data_ptr(), self.eps.data_ptr()
        f = self._fuse(k)
        y = self.gs.y
        std, gamma = self.gs._scalars
        f.y, f.y_sn = y.data_ptr(), self.y_sn
        f.p_start, f.p_step, f.p_stop = self.pos
        f.c_start, f.c_step, f.c_stop = self.chan
        f.std, f.gamma = std, gamma
        f.ghat = self.ghat.data_ptr()
        ops.net1d_launch_fused(d, f, False)

    def backward(self, mode: int, k: int, x: Optional[Tensor] = None, out: Optional[Tensor] = None):
        d, keep = self._desc(True, k)
        d.x = self.ghat.data_ptr()
        d.out = (self.out if out is None else out).data_ptr()
        f = self._fuse(k)
        f.eps = self.eps.data_ptr()
        f.mode = mode
        if mode == 1:
            f.xs, f.step_coef = x.data_ptr(), self.coef.data_ptr() + 16
        if mode == 2:
            f.partial, f.partial_stride = self.partial.data_ptr(), self.ptiles
        ops.net1d_launch_fused(d, f, True)

    # ------------------------------------------------------------------ one guided evaluation (GaussianScore.forward)
    def evaluate(self, x: Tensor, t: Tensor) -> Tensor:
        out = torch.empty_like(self.eps)
        self.prologue_eval(t)
        self.forward(x, 0)
        self.backward(0, 0, out=out)
        return out


class FusedLocal:
    """The same interface as :class:`Fused1D` for a LOCAL score network -- ``MCScoreNet`` over a ``ScoreNet`` kernel (sda/score.py:134-164,
    53-63; experiments/lorenz/utils.py:45-59): per evaluation ``sda_mlp_fwd_win`` (window gather + time embedding in the loader, ``fold`` + eps +
    likelihood cotangent in the epilogue), ``sda_mlp_bwd_win`` (fold's adjoint in the loader) and ``sda_mc_finish`` (overlapping-window sum, guided
    score, predictor update / Langevin sum of squares)."""

    def __init__(self, gs, inner, affine, mplan, x: Tensor, pos, chan, y_sn: int, sched):
        from . import mlp
        self.gs, self.inner, self.affine, self.mplan = gs, inner, affine, mplan
        self.pos, self.chan, self.y_sn, self.sched = pos, chan, y_sn, sched
        B, L, C = x.shape
        self.shape = (B, L, C)
        self.k = inner.order
        self.nw = L - 2 * self.k
        self.rows = B * self.nw
        dev = x.device
        self.eps = torch.empty(B, L, C, device=dev, dtype=torch.float32)
        self.ghat = torch.empty_like(self.eps)
        self.out = torch.empty_like(self.eps)
        self.gwin = torch.empty(self.rows, 16, device=dev, dtype=torch.float32)
        nres = max(mplan.nres, 1)
        self.a_s = torch.empty(nres, self.rows, mlp._MLP_W, device=dev, dtype=torch.float32)
        self.z_s = torch.empty_like(self.a_s)
        self.m_s = torch.empty(nres, self.rows, device=dev, dtype=torch.float32)
        self.r_s = torch.empty_like(self.m_s)
        self.coef = torch.zeros(COEF_LEN, device=dev, dtype=torch.float32)
        self.step_i = torch.zeros(1, device=dev, dtype=torch.int64)
        self.emb_n = inner.kernel.embedding[2].weight.shape[0]
        self.mod = torch.empty(2, self.emb_n, device=dev, dtype=torch.float32)        # the time embedding of both evaluation times
        self.ptiles = 1
        self.partial = torch.empty(B, 1, device=dev, dtype=torch.float32)

    def _prologue(self, table, istep, t, nt):
        emb = self.inner.kernel.embedding
        w0, b0, w2, b2 = emb[0].weight.detach(), emb[0].bias.detach(), emb[2].weight.detach(), emb[2].bias.detach()
        ak, eta, kk, sk = self.sched
        ops._dev(w0, b0, w2, b2, emb.freqs, t)
        _lib.check(_lib.load().sda_step1d_prologue(
            ops._ptr(table), 0 if table is None else table.shape[1], ops._ptr(istep), ops._ptr(t), nt, ak, eta, kk, sk,
            emb.freqs.data_ptr(), emb.freqs.numel(), w0.data_ptr(), b0.data_ptr(), w0.shape[0], w2.data_ptr(), b2.data_ptr(), w2.shape[0],
            None, None, 0, self.coef.data_ptr(), self.step_i.data_ptr(), self.mod.data_ptr(), ops._stream()), 'sda_step1d_prologue')

    def prologue_step(self, table: Tensor, istep: Tensor):
        self._prologue(table, istep, None, 2)

    def prologue_eval(self, t: Tensor):
        self._prologue(None, None, t.reshape(1), 1)

    def _desc(self, backward: bool):
        from . import mlp
        d = self.mplan.desc(self.rows, backward)
        d.a_save, d.z_save, d.save_stride, d.save_ld = self.a_s.data_ptr(), self.z_s.data_ptr(), self.rows * mlp._MLP_W, mlp._MLP_W
        d.mean_save, d.rstd_save, d.stat_stride = self.m_s.data_ptr(), self.r_s.data_ptr(), self.rows
        return d

    def _win(self, k: int):
        B, L, C = self.shape
        w = _lib.MlpWin()
        w.nw, w.len, w.c, w.emb_n = self.nw, L, C, self.emb_n
        w.cx0, w.cx1, w.cn = self.affine
        w.coef = self.coef.data_ptr() + 8 * k
        w.ghat = self.ghat.data_ptr()
        return w

    def forward(self, x: Tensor, k: int):
        d, w = self._desc(False), self._win(k)
        std, gamma = self.gs._scalars
        w.x, w.emb = x.data_ptr(), self.mod.data_ptr() + 4 * self.emb_n * k
        w.y, w.y_sn = self.gs.y.data_ptr(), self.y_sn
        w.p_start, w.p_step, w.p_stop = self.pos
        w.c_start, w.c_step, w.c_stop = self.chan
        w.std, w.gamma = std, gamma
        w.eps = self.eps.data_ptr()
        ops.mlp_launch_win(d, w, False)

    def backward(self, mode: int, k: int, x: Optional[Tensor] = None, out: Optional[Tensor] = None):
        B, L, C = self.shape
        d, w = self._desc(True), self._win(k)
        w.gwin = self.gwin.data_ptr()
        ops.mlp_launch_win(d, w, True)
        ops.mc_finish(self.eps, self.ghat, self.gwin, B, self.nw, self.k, C, self.affine[0], self.affine[1], self.coef.data_ptr() + 8 * k,
                      mode, None if mode == 1 else (self.out if out is None else out), x if mode == 1 else None,
                      self.coef.data_ptr() + 16, self.partial if mode == 2 else None)

    def evaluate(self, x: Tensor, t: Tensor) -> Tensor:
        out = torch.empty_like(self.eps)
        self.prologue_eval(t)
        self.forward(x, 0)
        self.backward(0, 0, out=out)
        return out


def _plan_local(gs, m, affine, x: Tensor, pos, chan, y_sn: int, sched, key_extra):
    """FusedLocal for ``m`` = MCScoreNet over a ScoreNet whose ResMLP the whole-MLP kernels take, or None."""
    from . import mlp
    from .score import ScoreNet
    kern = m.kernel
    if type(kern) is not ScoreNet or type(kern).forward is not ScoreNet.forward:
        return None
    B, L, C = x.shape
    k = m.order
    wc = (2 * k + 1) * C
    if L - 2 * k < 1 or wc > 16:
        return None
    mplan = mlp._fused_plan(list(kern.network))
    emb = kern.embedding
    if mplan is None or mplan.gemms[0][1] != wc + emb[2].weight.shape[0] or mplan.gemms[-1][2] != wc:
        return None
    if emb.freqs.numel() * 2 > 128 or emb[0].weight.shape[0] > 1024 or emb[2].weight.shape[0] > 256:
        return None
    key = key_extra + (id(mplan),)
    hit = getattr(gs, '_fused1d_cache', None)
    if hit is not None and hit[0] == key:
        return hit[1]
    fz = FusedLocal(gs, m, affine, mplan, x, pos, chan, y_sn, sched)
    object.__setattr__(gs, '_fused1d_cache', (key, fz))
    return fz


def plan(gs, x: Tensor, t, c):
    """The fused evaluation for this call of ``gs`` (a GaussianScore), or None when the general path must run."""
    from .observe import Subsample
    from .score import MCScoreNet, MCScoreWrapper, ScoreUNet
    if not ENABLED or gs.detach or c is not None or not torch.is_tensor(x) or not x.is_cuda or x.dtype != torch.float32 or x.dim() != 3 \
            or not x.is_contiguous() or x.shape[0] < 1:
        return None
    if t is not None and not (torch.is_tensor(t) and t.is_cuda and t.numel() == 1 and t.dtype == torch.float32):
        return None
    A = gs.A
    if not isinstance(A, Subsample) or type(A).__call__ is not Subsample.__call__ or not 1 <= len(A.slices) <= 2:
        return None
    if gs._scalars is None:
        return None
    sched = stock_schedule(gs.sde)
    if sched is None:
        return None
    m = gs.sde.eps
    affine = (0.0, 0.0, 1.0)
    form = getattr(m, 'affine_form', None)
    if form is not None:
        got = form(gs.sde)
        if got is None:
            return None
        m, affine = got[0], tuple(float(v) for v in got[1:])
    B, L, C = x.shape
    sl = A.slices
    pos = _slice3(sl[0], L) if len(sl) == 2 else (0, 1, L)
    chan = _slice3(sl[-1], C)
    if pos is None or chan is None:
        return None
    y = gs.y
    if not (torch.is_tensor(y) and y.is_cuda and y.dtype == torch.float32 and y.is_contiguous()):
        return None
    oshape = tuple(A._osize(x.shape))
    if tuple(y.shape) == oshape and B > 1:
        y_sn = oshape[1] * oshape[2]
    elif tuple(y.shape) == oshape[1:] or tuple(y.shape) == (1,) + oshape[1:] or tuple(y.shape) == oshape:
        y_sn = 0
    else:
        return None
    key0 = (tuple(x.shape), x.device, id(A), tuple((s_.start, s_.stop, s_.step) for s_ in sl), y.data_ptr(), tuple(y.shape), id(m), affine, sched)
    if type(m) is MCScoreNet:
        return _plan_local(gs, m, affine, x, pos, chan, y_sn, sched, key0)
    if type(m) is not MCScoreWrapper:
        return None
    score = m.score
    if not isinstance(score, ScoreUNet) or type(score).forward is not ScoreUNet.forward or type(score)._context is not ScoreUNet._context:
        return None
    unet = score.network
    if unet.spatial != 1 or unet.in_channels != C or unet.out_channels != C:
        return None
    engine = unet.engine()
    if engine.mod_total <= 0:
        return None
    src = Source(x=x, n=B, cx=C, hs=1, ws=L, sn_outer=L * C, sc=1, sy=0, sx=C)
    p1 = engine.net1d_plan(src)
    if p1 is None:
        return None
    emb = score.embedding
    if emb.freqs.numel() * 2 > 128 or emb[0].weight.shape[0] > 1024 or emb[2].weight.shape[0] > 256:
        return None
    key = key0 + (id(engine), len(p1['blocks']))
    hit = getattr(gs, '_fused1d_cache', None)
    if hit is not None and hit[0] == key:
        fz = hit[1]
        fz.plan1d = p1
        return fz
    fz = Fused1D(gs, m, affine, engine, p1, x, pos, chan, y_sn, sched)
    object.__setattr__(gs, '_fused1d_cache', (key, fz))
    return fz
