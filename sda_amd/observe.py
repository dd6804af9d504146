r"""Observation operators with hand-written adjoints / linearised VJPs.

``GaussianScore`` accepts any callable ``A`` and differentiates it with torch autograd, exactly as the reference.
If ``A`` is one of the operators below (anything exposing ``linearize``), the guidance gradient is formed without
autograd: :math:`\nabla_{\hat x} \log p = J_A(\hat x)^T\big((y - A\hat x)/\mathrm{var}\big)` -- streaming HIP kernels,
hipGraph-capturable.  The catalogue covers every ``A`` of the reference's experiments:

    A = Subsample((slice(None, None, 8), slice(0, 1)))                             # x[..., ::8, :1]   lorenz/eval.py:75
    A = Subsample.space(4)                                                         # x[..., ::4, ::4]  figures.ipynb#cell33
    A = Compose(Subsample.frames(4), Coarsen(8))                                   # figures.ipynb#cell9
    A = Compose(Coarsen(4), Crop((slice(None, None, 3), slice(None), slice(4, 12), slice(4, 12))))     # #cell16
    A = Compose(Subsample.frames(3), Coarsen(4), Vorticity(), Pointwise('saturate'),
                Crop((slice(2, 14), slice(2, 14))))                                # #cell23: w / (1 + |w|), cropped
    A = Compose(Select(-4, -1), Vorticity(), Mask(mask))                           # #cell4: vorticity of the last frame x mask
    A = TimeDiff(dim=1, i=0, j=-1)                                                 # #cell43: loop closure x[:, 0] - x[:, -1]
"""
import ctypes
from typing import Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import _lib, ops

_I5 = ctypes.c_int * 5


def _stream():
    return torch.cuda.current_stream().cuda_stream


class Observation:
    """An observation operator with the VJP of its linearisation; instances are callables usable wherever the reference
    takes ``A`` (sda/score.py:356-364).  ``linearize(x) -> (A(x), r -> J_A(x)^T r)`` is what GaussianScore calls."""

    def __call__(self, x: Tensor) -> Tensor:
        raise NotImplementedError

    def linearize(self, x: Tensor):
        raise NotImplementedError

    def out_shape(self, x_shape) -> tuple:
        """Shape of ``A(x)`` for an input of shape ``x_shape`` (no launch)."""
        raise NotImplementedError

    def __rshift__(self, other: 'Observation') -> 'Compose':
        return Compose(self, other)


class LinearObservation(Observation):
    """A linear map with an adjoint (``adjoint(r, x_shape) = A^T r``): its linearisation does not depend on ``x``."""

    def adjoint(self, r: Tensor, x_shape) -> Tensor:
        raise NotImplementedError

    def linearize(self, x: Tensor):
        shape = tuple(x.shape)
        return self(x), (lambda r: self.adjoint(r, shape))


class Subsample(LinearObservation):
    r"""``x[..., s_k]`` with one ``slice(start, None, step)`` per trailing dim, e.g. ``Subsample((slice(None, None, 8),
    slice(0, 1)))`` = ``x[..., ::8, :1]`` (lorenz/eval.py:75) or ``Subsample.space(4)`` = ``x[..., ::4, ::4]``."""

    def __init__(self, slices: Sequence[slice]):
        self.slices = tuple(slices)
        if len(self.slices) > 5:
            raise ValueError('at most 5 trailing dims')

    @classmethod
    def space(cls, step: int, offset: int = 0) -> 'Subsample':
        return cls((slice(offset, None, step), slice(offset, None, step)))

    @classmethod
    def frames(cls, step: int, offset: int = 0) -> 'Subsample':
        """``x[..., offset::step, :, :, :]``: every ``step``-th frame of a (..., L, C, H, W) trajectory."""
        return cls((slice(offset, None, step), slice(None), slice(None), slice(None)))

    def _spec(self, shape):
        nd = len(self.slices)
        lead = 1
        for s in shape[:len(shape) - nd]:
            lead *= s
        tail = list(shape[len(shape) - nd:])
        size, start, step, stop_ok = [lead] + tail, [0], [1], []
        for sl, n in zip(self.slices, tail):
            a, b, c = sl.indices(n)
            if c <= 0:
                raise ValueError('negative steps are not supported')
            if b < n:                       # a stop: express as a shorter extent
                n_eff = b
            else:
                n_eff = n
            start.append(a); step.append(c); stop_ok.append(n_eff)
        # fold leading dims so that there are exactly 5
        while len(size) < 5:
            size.insert(0, 1); start.insert(0, 0); step.insert(0, 1)
        if len(size) > 5:
            raise ValueError('too many dims')
        return size, start, step, stop_ok

    def _osize(self, shape):
        nd = len(self.slices)
        out = list(shape[:len(shape) - nd])
        for sl, n in zip(self.slices, shape[len(shape) - nd:]):
            out.append(len(range(*sl.indices(n))))
        return out

    def _stop(self, size, stops):
        stop = list(size)
        stop[5 - len(stops):] = stops                      # (exclusive ends of the sliced trailing dims)
        return stop

    def out_shape(self, x_shape) -> tuple:
        return tuple(self._osize(tuple(x_shape)))

    def __call__(self, x: Tensor) -> Tensor:
        ops._dev(x)
        xs = x.contiguous()
        size, start, step, stops = self._spec(xs.shape)
        out = torch.empty(self._osize(xs.shape), device=x.device, dtype=torch.float32)
        if out.numel() == 0:
            return out
        _lib.check(_lib.load().sda_obs_subsample(xs.data_ptr(), _I5(*size), _I5(*start), _I5(*step), _I5(*self._stop(size, stops)),
                                                 out.data_ptr(), _stream()), 'sda_obs_subsample')
        return out

    def gaussian_guidance(self, x: Tensor, eps: Tensor, y: Tensor, std: float, gamma: float, mu, sigma) -> Optional[Tensor]:
        """``A^T((y - A((x - sigma eps)/mu)) / (std^2 + gamma (sigma/mu)^2))`` in one launch, or None when this operator /
        these shapes need the general path (y not broadcastable over the leading axis, other dtypes)."""
        ops._dev(x, eps, y)
        oshape = tuple(self._osize(x.shape))
        if not (tuple(y.shape) == oshape or tuple(y.shape) == oshape[1:] or
                (y.dim() == len(oshape) and y.shape[0] == 1 and tuple(y.shape[1:]) == oshape[1:])):
            return None
        if x.dtype != torch.float32 or eps.dtype != torch.float32 or y.dtype != torch.float32:
            return None
        xs, es, ys = x.contiguous(), eps.contiguous(), y.contiguous()
        size, start, step, stops = self._spec(xs.shape)
        stop = self._stop(size, stops)
        g = torch.empty_like(xs)
        m, s_, pair = ops._coef(mu, sigma)
        _lib.check(_lib.load().sda_obs_subsample_guidance(xs.data_ptr(), es.data_ptr(), ys.data_ptr(), ys.numel(), _I5(*size),
                                                          _I5(*start), _I5(*step), _I5(*stop), float(std), float(gamma), m, s_,
                                                          ops._ptr(pair), g.data_ptr(), _stream()), 'sda_obs_subsample_guidance')
        return g

    def adjoint(self, r: Tensor, x_shape) -> Tensor:
        ops._dev(r)
        x_shape = tuple(x_shape)
        size, start, step, stops = self._spec(x_shape)
        gx = torch.empty(x_shape, device=r.device, dtype=torch.float32)
        _lib.check(_lib.load().sda_obs_subsample_adjoint(r.contiguous().data_ptr(), _I5(*size), _I5(*start), _I5(*step),
                                                         _I5(*self._stop(size, stops)), gx.data_ptr(), _stream()),
                   'sda_obs_subsample_adjoint')
        return gx


class Crop(Subsample):
    r"""``x[..., a:b, c:d]`` -- a :class:`Subsample` whose slices carry stops (figures.ipynb#cell16, #cell23); same kernels."""


class Select(LinearObservation):
    r"""``x.select(dim, index)`` (drops the axis), e.g. ``Select(-4, -1)`` = ``x[..., -1, :, :, :]`` (figures.ipynb#cell4)."""

    def __init__(self, dim: int, index: int):
        if dim >= 0:
            raise ValueError('Select counts dims from the end (negative dim), like the reference\'s Ellipsis indexing')
        self.dim, self.index = dim, index

    def _sub(self, x_shape):
        n = x_shape[self.dim]
        i = self.index % n
        return Subsample((slice(i, i + 1),) + (slice(None),) * (-self.dim - 1))

    def out_shape(self, x_shape) -> tuple:
        x_shape = tuple(x_shape)
        k = len(x_shape) + self.dim
        return x_shape[:k] + x_shape[k + 1:]

    def __call__(self, x: Tensor) -> Tensor:
        return self._sub(x.shape)(x).squeeze(self.dim)

    def adjoint(self, r: Tensor, x_shape) -> Tensor:
        return self._sub(tuple(x_shape)).adjoint(r.unsqueeze(self.dim), x_shape)


class Coarsen(LinearObservation):
    r"""Block mean over ``f x f`` cells of the last two dims (``KolmogorovFlow.coarsen``, mcs.py:340-347)."""

    def __init__(self, f: int = 2):
        self.f = f

    def out_shape(self, x_shape) -> tuple:
        *lead, h, w = tuple(x_shape)
        return (*lead, h // self.f, w // self.f)

    def __call__(self, x: Tensor) -> Tensor:
        ops._dev(x)
        xs = x.contiguous()
        *lead, h, w = xs.shape
        planes = 1
        for s in lead:
            planes *= s
        out = torch.empty(*lead, h // self.f, w // self.f, device=x.device, dtype=torch.float32)
        _lib.check(_lib.load().sda_obs_coarsen(xs.data_ptr(), planes, h, w, self.f, out.data_ptr(), _stream()),
                   'sda_obs_coarsen')
        return out

    def adjoint(self, r: Tensor, x_shape) -> Tensor:
        ops._dev(r)
        *lead, h, w = tuple(x_shape)
        planes = 1
        for s in lead:
            planes *= s
        gx = torch.empty(tuple(x_shape), device=r.device, dtype=torch.float32)
        _lib.check(_lib.load().sda_obs_coarsen_adjoint(r.contiguous().data_ptr(), planes, h, w, self.f, gx.data_ptr(),
                                                       _stream()), 'sda_obs_coarsen_adjoint')
        return gx


class Vorticity(LinearObservation):
    r"""``(..., 2, H, W) -> (..., H, W)``: :math:`\partial_x u - \partial_y v` by periodic central differences
    (``KolmogorovFlow.vorticity``, mcs.py:361-375)."""

    def out_shape(self, x_shape) -> tuple:
        *lead, two, h, w = tuple(x_shape)
        return (*lead, h, w)

    def __call__(self, x: Tensor) -> Tensor:
        ops._dev(x)
        xs = x.contiguous()
        *lead, two, h, w = xs.shape
        assert two == 2
        pairs = 1
        for s in lead:
            pairs *= s
        out = torch.empty(*lead, h, w, device=x.device, dtype=torch.float32)
        _lib.check(_lib.load().sda_obs_vorticity(xs.data_ptr(), pairs, h, w, out.data_ptr(), _stream()), 'sda_obs_vorticity')
        return out

    def adjoint(self, r: Tensor, x_shape) -> Tensor:
        ops._dev(r)
        *lead, two, h, w = tuple(x_shape)
        pairs = 1
        for s in lead:
            pairs *= s
        gx = torch.empty(tuple(x_shape), device=r.device, dtype=torch.float32)
        _lib.check(_lib.load().sda_obs_vorticity_adjoint(r.contiguous().data_ptr(), pairs, h, w, gx.data_ptr(), _stream()),
                   'sda_obs_vorticity_adjoint')
        return gx


_PW_KINDS = {'saturate': 1, 'tanh': 2, 'square': 3, 'abs': 4}


class Pointwise(Observation):
    r"""An elementwise non-linearity ``f`` with derivative ``df``; VJP of the linearisation: ``r * df(x)``.

    ``Pointwise('saturate')`` is :math:`w / (1 + |w|)` (the saturating sensor of figures.ipynb#cell23); also ``'tanh'``,
    ``'square'``, ``'abs'`` -- one HIP launch each way.  ``Pointwise(f, df)`` takes two torch callables for anything else
    (``df`` is evaluated explicitly: no autograd graph is built either way)."""

    def __init__(self, f, df=None):
        if isinstance(f, str):
            if f not in _PW_KINDS:
                raise ValueError(f'unknown pointwise function {f!r} (known: {sorted(_PW_KINDS)})')
            self.kind, self.f, self.df = _PW_KINDS[f], None, None
        else:
            if df is None:
                raise ValueError('Pointwise(f, df): the derivative is required (no autograd through A)')
            self.kind, self.f, self.df = 0, f, df

    def out_shape(self, x_shape) -> tuple:
        return tuple(x_shape)

    def __call__(self, x: Tensor) -> Tensor:
        ops._dev(x)
        if not self.kind:
            with torch.no_grad():
                return self.f(x)
        xs = x.contiguous()
        out = torch.empty_like(xs)
        _lib.check(_lib.load().sda_obs_pointwise(xs.data_ptr(), xs.numel(), self.kind, out.data_ptr(), _stream()), 'sda_obs_pointwise')
        return out

    def linearize(self, x: Tensor):
        xs = x.contiguous()

        def vjp(r: Tensor) -> Tensor:
            ops._dev(r)
            if not self.kind:
                with torch.no_grad():
                    return r * self.df(xs)
            gx = torch.empty_like(xs)
            _lib.check(_lib.load().sda_obs_pointwise_vjp(xs.data_ptr(), r.contiguous().data_ptr(), xs.numel(), self.kind,
                                                         gx.data_ptr(), _stream()), 'sda_obs_pointwise_vjp')
            return gx
        return self(xs), vjp


class Mask(LinearObservation):
    r"""``x * mask`` with the mask broadcast over the leading dims (figures.ipynb#cell4: a boolean annulus); self-adjoint."""

    def __init__(self, mask: Tensor):
        self.mask = mask

    def _m(self, device) -> Tensor:
        m = self.mask
        if m.dtype != torch.float32 or m.device != device or not m.is_contiguous():
            m = self.mask = m.to(device=device, dtype=torch.float32).contiguous()
        return m

    def out_shape(self, x_shape) -> tuple:
        return tuple(x_shape)

    def __call__(self, x: Tensor) -> Tensor:
        ops._dev(x)
        xs = x.contiguous()
        m = self._m(xs.device)
        if tuple(xs.shape[xs.dim() - m.dim():]) != tuple(m.shape):
            raise _lib.SdaHipError(f'Mask: mask shape {tuple(m.shape)} does not match the trailing dims of {tuple(xs.shape)}')
        out = torch.empty_like(xs)
        _lib.check(_lib.load().sda_obs_mask(xs.data_ptr(), xs.numel(), m.data_ptr(), m.numel(), out.data_ptr(), _stream()),
                   'sda_obs_mask')
        return out

    def adjoint(self, r: Tensor, x_shape) -> Tensor:
        return self(r)


class TimeDiff(LinearObservation):
    r"""``x.select(dim, i) - x.select(dim, j)``: the loop-closure constraint ``x[:, 0] - x[:, -1]`` of figures.ipynb#cell43
    (``dim`` counts from the front, as there)."""

    def __init__(self, dim: int = 1, i: int = 0, j: int = -1):
        self.dim, self.i, self.j = dim, i, j

    def _geom(self, x_shape):
        x_shape = tuple(x_shape)
        d = self.dim % len(x_shape)
        outer = inner = 1
        for s_ in x_shape[:d]:
            outer *= s_
        for s_ in x_shape[d + 1:]:
            inner *= s_
        n = x_shape[d]
        return d, outer, n, inner, self.i % n, self.j % n

    def out_shape(self, x_shape) -> tuple:
        x_shape = tuple(x_shape)
        d = self.dim % len(x_shape)
        return x_shape[:d] + x_shape[d + 1:]

    def __call__(self, x: Tensor) -> Tensor:
        ops._dev(x)
        xs = x.contiguous()
        d, outer, n, inner, i, j = self._geom(xs.shape)
        out = torch.empty(self.out_shape(xs.shape), device=x.device, dtype=torch.float32)
        _lib.check(_lib.load().sda_obs_timediff(xs.data_ptr(), outer, n, inner, i, j, out.data_ptr(), _stream()), 'sda_obs_timediff')
        return out

    def adjoint(self, r: Tensor, x_shape) -> Tensor:
        ops._dev(r)
        d, outer, n, inner, i, j = self._geom(x_shape)
        gx = torch.empty(tuple(x_shape), device=r.device, dtype=torch.float32)
        _lib.check(_lib.load().sda_obs_timediff_adjoint(r.contiguous().data_ptr(), outer, n, inner, i, j, gx.data_ptr(), _stream()),
                   'sda_obs_timediff_adjoint')
        return gx


class Compose(Observation):
    """``Compose(A1, A2, ...)(x) = ...A2(A1(x))``; the VJP applies the members' VJPs in reverse (each linearised at its own
    input).  A composition of linear members is linear and also offers ``adjoint``."""

    def __init__(self, *ops_: Observation):
        self.ops = ops_

    def __call__(self, x: Tensor) -> Tensor:
        for op in self.ops:
            x = op(x)
        return x

    def out_shape(self, x_shape) -> tuple:
        shape = tuple(x_shape)
        for op in self.ops:
            shape = op.out_shape(shape)
        return shape

    def linearize(self, x: Tensor):
        vjps = []
        for op in self.ops:
            x, v = op.linearize(x)
            vjps.append(v)

        def vjp(r: Tensor) -> Tensor:
            for v in reversed(vjps):
                r = v(r)
            return r
        return x, vjp

    def adjoint(self, r: Tensor, x_shape) -> Tensor:
        if not all(isinstance(op, LinearObservation) for op in self.ops):
            raise _lib.SdaHipError('Compose.adjoint: a non-linear member has no fixed adjoint (use linearize)')
        shapes = [tuple(x_shape)]
        for op in self.ops[:-1]:
            shapes.append(op.out_shape(shapes[-1]))
        for op, shp in zip(reversed(self.ops), reversed(shapes)):
            r = op.adjoint(r, shp)
        return r
