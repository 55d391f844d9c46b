"""TEST INFRASTRUCTURE (checker only; never imported by the product path).

CPU restatement, in numpy, of the counter-based noise that `lsnf_langevin_step` can draw inside its update
kernel (include/lsnf_flow.h `LsnfRng`; device code csrc/lsnf_device.h `lsnf_noise_tile`).  It stands in for
`torch.randn_like(z)` of the reference's Langevin step (train.py:326).  The reference's noisy sampler is pinned
only statistically (RNG streams differ across devices); this oracle pins OUR stream exactly:

  Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; the Random123 known-answer
  vectors are checked in tests/test_oracle_golden.py), Box-Muller on 24-bit uniforms.

  noise(row, col), col = hh*half + f:  normal number (f & 3) of the block
      counter = ((hh << 16) | (f >> 2), row & 0xffffffff, offset & 0xffffffff, (offset >> 32) ^ (row >> 32))
      key     = (seed & 0xffffffff, seed >> 32)
"""
import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  All arguments broadcastable integer arrays (values < 2**32); returns 4 uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & _MASK for c in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = _M0 * c0, _M1 * c2                      # 32x32 -> 64 bit products
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & _MASK, p1 >> np.uint64(32), p1 & _MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def _box_muller(xa, xb):
    u1 = ((xa >> np.uint32(8)).astype(np.float64) + 0.5) * 2.0 ** -24
    u2 = ((xb >> np.uint32(8)).astype(np.float64) + 0.5) * 2.0 ** -24
    r = np.sqrt(-2.0 * np.log(u1))
    return r * np.cos(2.0 * np.pi * u2), r * np.sin(2.0 * np.pi * u2)


def langevin_noise(B, nz, seed, offset=0, row0=0):
    """(B, nz) float64 array of the N(0,1) draws the kernel makes for rows row0 .. row0+B-1."""
    half = nz // 2
    rows = (np.arange(B, dtype=np.uint64) + np.uint64(row0))[:, None]
    cols = np.arange(nz)
    hh, f = cols // half, cols % half
    c0 = ((hh << 16) | (f >> 2))[None, :]
    c1 = rows & _MASK
    c2 = np.uint64(offset & 0xFFFFFFFF)
    c3 = np.uint64((offset >> 32) & 0xFFFFFFFF) ^ (rows >> np.uint64(32))
    x0, x1, x2, x3 = philox4x32_10(c0, c1, c2, c3, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    n0, n1 = _box_muller(x0, x1)
    n2, n3 = _box_muller(x2, x3)
    sel = (f & 3)[None, :]
    return np.where(sel == 0, n0, np.where(sel == 1, n1, np.where(sel == 2, n2, n3)))
