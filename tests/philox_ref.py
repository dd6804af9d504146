"""TEST-ONLY numpy restatement of the row-keyed normal generator of sda_amd/csrc/noise.hip (Philox4x32-10, Salmon et al.
SC'11, counter {quad, row, draw_lo, draw_hi}, key = seed; Box-Muller on 24-bit uniforms).  The GPU tests compare the
kernel's raw words bit-for-bit and its normals to float round-off against this file; the CPU multi-process tests use it as
the stand-in for the device kernel."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xffffffff)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """uint32 arrays (broadcastable) -> four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(v, dtype=np.uint64) & MASK for v in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0, k1 = int(k0) & 0xffffffff, int(k1) & 0xffffffff
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0, k1 = (k0 + W0) & 0xffffffff, (k1 + W1) & 0xffffffff
    return tuple(v.astype(np.uint32) for v in (c0, c1, c2, c3))


def philox_words(n, seed, c1, c2, c3):
    w = philox4x32_10(np.arange(n, dtype=np.uint64), c1, c2, c3, seed & 0xffffffff, (seed >> 32) & 0xffffffff)
    return np.stack(w, axis=1).reshape(-1)


def _box_muller(a, b):
    u1 = ((a >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
    u2 = ((b >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
    r = np.sqrt(np.float32(-2.0) * np.log(u1.astype(np.float64))).astype(np.float64)
    ang = np.float64(np.float32(6.28318530717958647692)) * u2.astype(np.float64)
    ang = ang.astype(np.float32).astype(np.float64)          # the device forms the angle in fp32
    return (r * np.cos(ang)).astype(np.float32), (r * np.sin(ang)).astype(np.float32)


def randn_rows(rows, per_row, seed, row0, draw):
    """(rows, per_row) float32: row r, element j <- (seed, row0 + r, draw, j)."""
    quads = (per_row + 3) // 4
    q = np.arange(quads, dtype=np.uint64)[None, :]
    grow = (np.arange(rows, dtype=np.uint64) + np.uint64(row0))[:, None]
    draw = int(draw) & 0xffffffffffffffff
    c3 = (np.uint64(draw >> 32) ^ ((q >> np.uint64(32)) << np.uint64(16)) ^ ((grow >> np.uint64(32)) << np.uint64(24))) & MASK
    x0, x1, x2, x3 = philox4x32_10(q & MASK, grow & MASK, np.uint64(draw & 0xffffffff), c3, seed & 0xffffffff,
                                   (seed >> 32) & 0xffffffff)
    z0, z1 = _box_muller(x0, x1)
    z2, z3 = _box_muller(x2, x3)
    out = np.stack((z0, z1, z2, z3), axis=-1).reshape(rows, quads * 4)
    return out[:, :per_row]
