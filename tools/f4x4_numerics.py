#!/usr/bin/env python3
"""F(4x4,3x3) vs F(2x2,3x3) vs direct convolution in fp32 against float64 (CPU, torch; Cook-Toom matrices built with sympy):
the numerics half of the go / no-go in profiles/r03_f4x4_feasibility.txt.

    python tools/f4x4_numerics.py
"""
import torch, math
torch.manual_seed(0)
def mats(points):
    # Cook-Toom construction for F(m, r) with m + r - 1 = len(points) + 1 (last point = infinity), in float64
    import numpy as np
    from fractions import Fraction
    n = len(points) + 1; r = 3; m = n - r + 1
    pts = [Fraction(p) for p in points]
    # Vandermonde-based: A^T (m x n), G (n x r), B^T (n x n) such that Y = A^T[(G g) . (B^T d)]
    # Use the standard construction via polynomial interpolation (wincnn style)
    import sympy as sp
    a = [sp.Rational(p.numerator, p.denominator) for p in pts]
    x = sp.symbols('x')
    def At(a, m, n):
        return sp.Matrix(m, n, lambda i, j: a[j] ** i if j < n - 1 else (1 if i == m - 1 else 0))
    def fdiag(a):
        n = len(a)
        f = []
        for i in range(n):
            p = 1
            for j in range(n):
                if j != i: p *= (a[i] - a[j])
            f.append(p)
        return f
    f = fdiag(a)
    AT = At(a, m, n)
    G = sp.Matrix(n, r, lambda i, j: (a[i] ** j / f[i]) if i < n - 1 else (1 if j == r - 1 else 0))
    # B^T from Lagrange polynomials
    M = sp.prod([(x - ai) for ai in a])
    BT = sp.zeros(n, n)
    for i in range(n - 1):
        Li = sp.Poly(sp.expand(sp.cancel(M / (x - a[i]))), x)   # prod_{j != i} (x - a_j), degree n-2
        co = Li.all_coeffs()[::-1]
        for j in range(len(co)): BT[i, j] = co[j]
    co = sp.Poly(sp.expand(M), x).all_coeffs()[::-1]
    for j in range(len(co)): BT[n - 1, j] = co[j]
    # scale: Y = A^T [ (G g) . (B^T d) ] holds with f folded in G (done above)
    tof = lambda Mx: torch.tensor([[float(v) for v in Mx.row(i)] for i in range(Mx.rows)], dtype=torch.float64)
    return tof(AT), tof(G), tof(BT)

def wino_conv(x, w, AT, G, BT, dtype):
    # x: (N, C, H, W) circular pad; w: (K, C, 3, 3); tiles of m x m outputs
    m = AT.shape[0]; n = BT.shape[0]
    N, C, H, W = x.shape
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1), mode='circular')
    # extract patches n x n with stride m
    pat = xp.unfold(2, n, m).unfold(3, n, m)             # N, C, th, tw, n, n
    BTd, Gd, ATd = BT.to(dtype), G.to(dtype), AT.to(dtype)
    V = torch.einsum('ij,nctujk,lk->nctuil', BTd, pat.to(dtype), BTd)     # B^T d B
    U = torch.einsum('ij,kcjl,ml->kcim', Gd, w.to(dtype), Gd)             # G g G^T  (K, C, n, n)
    Mm = torch.einsum('kcil,nctuil->nktuil', U, V)                        # sum over c -- accumulate in dtype
    Y = torch.einsum('ij,nktujl,ml->nktuim', ATd, Mm, ATd)                # N,K,th,tw,m,m
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, w.shape[0], H, W)

def ref(x, w):
    return torch.nn.functional.conv2d(torch.nn.functional.pad(x.double(), (1,1,1,1), mode='circular'), w.double())

def rel(a, b): return ((a.double() - b).abs().max() / b.abs().max()).item()
def rms(a, b): return ((a.double() - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()

F22 = mats([0, 1, -1])
F44 = mats([0, 1, -1, 2, -2])
F44h = mats([0, 1, -1, sp_half := 0.5, -0.5]) if False else None
from fractions import Fraction as Fr
F44h = mats([0, 1, -1, Fr(1,2), Fr(-1,2)])
F33 = mats([0, 1, -1, 2])    # F(3x3,3x3): 5 points+inf... n=5
for C, K, S in ((96, 96, 48), (192, 192, 24), (384, 384, 24)):
    x = torch.randn(2, C, S, S)                       # LayerNorm-ed activations are ~unit variance
    w = (torch.rand(K, C, 3, 3) * 2 - 1) / math.sqrt(C * 9)   # torch default conv init range
    r = ref(x, w)
    d32 = torch.nn.functional.conv2d(torch.nn.functional.pad(x, (1,1,1,1), mode='circular'), w)
    out = [f'C={C}: direct fp32 max {rel(d32, r):.2e} rms {rms(d32, r):.2e}']
    for name, Ms in (('F(2x2)', F22), ('F(4x4) pts 0,+-1,+-2', F44), ('F(4x4) pts 0,+-1,+-1/2', F44h)):
        AT, G, BT = Ms
        m = AT.shape[0]
        if S % m: continue
        y64 = wino_conv(x, w, AT, G, BT, torch.float64)
        assert rel(y64, r) < 1e-10, (name, rel(y64, r))
        y32 = wino_conv(x, w, AT, G, BT, torch.float32)
        out.append(f'{name}: max {rel(y32, r):.2e} rms {rms(y32, r):.2e}')
    print(' | '.join(out), flush=True)

print('--- chain: 18 modulated residual blocks x + conv2(silu(conv1(LN(x)))) at C=96, 24x24, then the VJP chain back (fp64 autograd of the fp64 net as reference)')
def ln(x): 
    v, m = torch.var_mean(x, dim=1, unbiased=True, keepdim=True); return (x - m) / torch.sqrt(v + 1e-5)
def chain(x, ws, conv):
    for w1, w2 in ws:
        x = x + conv(torch.nn.functional.silu(conv(ln(x), w1)), w2)
    return x
C, S = 96, 24
ws = [((torch.rand(C, C, 3, 3) * 2 - 1) / math.sqrt(C * 9), (torch.rand(C, C, 3, 3) * 2 - 1) / math.sqrt(C * 9)) for _ in range(18)]
x0 = torch.randn(2, C, S, S)
r = chain(x0.double(), [(a.double(), b.double()) for a, b in ws], lambda a, w: ref(a, w))
for name, Ms in (('direct fp32', None), ('F(2x2)', F22), ('F(4x4) pts 0,+-1,+-2', F44), ('F(4x4) pts 0,+-1,+-1/2', F44h)):
    if Ms is None:
        conv = lambda a, w: torch.nn.functional.conv2d(torch.nn.functional.pad(a, (1,1,1,1), mode='circular'), w)
    else:
        conv = lambda a, w, Ms=Ms: wino_conv(a, w, *Ms, torch.float32)
    y = chain(x0, ws, conv)
    print(f'{name:26s} forward chain: max {rel(y, r):.2e} rms {rms(y, r):.2e}', flush=True)
