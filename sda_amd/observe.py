r"""Linear observation operators with hand-written adjoints.

``GaussianScore`` accepts any callable ``A`` and differentiates it with torch autograd, exactly as the reference.
If ``A`` is one of the operators below (anything exposing ``adjoint``), the guidance gradient is formed without
autograd: :math:`\nabla_{\hat x} \log p = A^T\big((y - A\hat x)/\mathrm{var}\big)` -- two streaming HIP kernels.

    A = Subsample(time=4) >> ... ; or  A = Compose(Subsample(space=4))          # x[..., ::4, ::4]
    A = Compose(Subsample(time=4), Coarsen(8))                                   # kolmogorov/figures.ipynb#cell9
"""
import ctypes
from typing import Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import _lib, ops

_I5 = ctypes.c_int * 5


def _stream():
    return torch.cuda.current_stream().cuda_stream


class LinearObservation:
    """A linear map with an adjoint; instances are callables usable wherever the reference takes ``A``."""

    def __call__(self, x: Tensor) -> Tensor:
        raise NotImplementedError

    def adjoint(self, r: Tensor, x_shape) -> Tensor:
        raise NotImplementedError

    def __rshift__(self, other: 'LinearObservation') -> 'Compose':
        return Compose(self, other)


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

    def __call__(self, x: Tensor) -> Tensor:
        ops._dev(x)
        xs = x.contiguous()
        size, start, step, _ = self._spec(xs.shape)
        oshape = self._osize(xs.shape)
        # stops shorter than the dim: gather over the truncated extent by shrinking `size` is wrong for strides, so
        # handle stops by limiting the output size explicitly: the kernel derives osize from size/start/step, hence
        # we pass a size vector whose sliced dims are cut at the stop and add the row pitch through a copy only if needed
        if any(sl.indices(n)[1] < n for sl, n in zip(self.slices, xs.shape[len(xs.shape) - len(self.slices):])):
            xs = xs[(Ellipsis,) + tuple(slice(0, sl.indices(n)[1]) for sl, n in
                                         zip(self.slices, xs.shape[len(xs.shape) - len(self.slices):]))].contiguous()
            size, start, step, _ = self._spec(xs.shape)
        out = torch.empty(oshape, device=x.device, dtype=torch.float32)
        _lib.check(_lib.load().sda_obs_subsample(xs.data_ptr(), _I5(*size), _I5(*start), _I5(*step), out.data_ptr(),
                                                 _stream()), 'sda_obs_subsample')
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
        stop = list(size)
        stop[5 - len(stops):] = stops                      # (exclusive ends of the sliced trailing dims)
        g = torch.empty_like(xs)
        m, s_, pair = ops._coef(mu, sigma)
        _lib.check(_lib.load().sda_obs_subsample_guidance(xs.data_ptr(), es.data_ptr(), ys.data_ptr(), ys.numel(), _I5(*size),
                                                          _I5(*start), _I5(*step), _I5(*stop), float(std), float(gamma), m, s_,
                                                          ops._ptr(pair), g.data_ptr(), _stream()), 'sda_obs_subsample_guidance')
        return g

    def adjoint(self, r: Tensor, x_shape) -> Tensor:
        ops._dev(r)
        x_shape = tuple(x_shape)
        nd = len(self.slices)
        tail = x_shape[len(x_shape) - nd:]
        cut = tuple(sl.indices(n)[1] for sl, n in zip(self.slices, tail))
        if any(c < n for c, n in zip(cut, tail)):
            inner_shape = x_shape[:len(x_shape) - nd] + cut
            inner = self._adjoint_full(r, inner_shape)
            gx = torch.zeros(x_shape, device=r.device, dtype=torch.float32)
            gx[(Ellipsis,) + tuple(slice(0, c) for c in cut)] = inner
            return gx
        return self._adjoint_full(r, x_shape)

    def _adjoint_full(self, r: Tensor, x_shape) -> Tensor:
        size, start, step, _ = self._spec(x_shape)
        gx = torch.empty(x_shape, device=r.device, dtype=torch.float32)
        _lib.check(_lib.load().sda_obs_subsample_adjoint(r.contiguous().data_ptr(), _I5(*size), _I5(*start), _I5(*step),
                                                         gx.data_ptr(), _stream()), 'sda_obs_subsample_adjoint')
        return gx


class Coarsen(LinearObservation):
    r"""Block mean over ``f x f`` cells of the last two dims (``KolmogorovFlow.coarsen``, mcs.py:340-347)."""

    def __init__(self, f: int = 2):
        self.f = f

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


class Compose(LinearObservation):
    """``Compose(A1, A2, ...)(x) = ...A2(A1(x))``; adjoint applies the adjoints in reverse."""

    def __init__(self, *ops_: LinearObservation):
        self.ops = ops_

    def __call__(self, x: Tensor) -> Tensor:
        for op in self.ops:
            x = op(x)
        return x

    def adjoint(self, r: Tensor, x_shape) -> Tensor:
        shapes = [tuple(x_shape)]
        probe = torch.empty(x_shape, device='meta')
        for op in self.ops[:-1]:
            probe = op.meta(probe) if hasattr(op, 'meta') else _meta_apply(op, probe)
            shapes.append(tuple(probe.shape))
        for op, shp in zip(reversed(self.ops), reversed(shapes)):
            r = op.adjoint(r, shp)
        return r


def _meta_apply(op: LinearObservation, probe: Tensor) -> Tensor:
    if isinstance(op, Subsample):
        return torch.empty(op._osize(probe.shape), device='meta')
    if isinstance(op, Coarsen):
        *lead, h, w = probe.shape
        return torch.empty(*lead, h // op.f, w // op.f, device='meta')
    if isinstance(op, Vorticity):
        *lead, two, h, w = probe.shape
        return torch.empty(*lead, h, w, device='meta')
    raise NotImplementedError(type(op))
