"""ORACLE — TEST INFRASTRUCTURE ONLY.

Host (numpy) statement of the counter-based Gaussian generator the engine uses when no noise
tensors are injected (`kd_philox_normal`, kernels_sampler.hip).  The reference itself is unseeded
(no seed is set anywhere), so this contract is the build's own:

  Philox4x32-10, key = (seed_lo, seed_hi), counter = (i4_lo, i4_hi, stream_lo, stream_hi) with
  i4 = element_index // 4; the four 32-bit outputs give two Box-Muller pairs
  u = ((x >> 8) + 0.5) * 2^-24,  z0 = sqrt(-2 ln u1) cos(2 pi u2),  z1 = sqrt(-2 ln u1) sin(2 pi u2).
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & MASK for c in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def philox_normal(n: int, seed: int, stream_id: int) -> np.ndarray:
    n4 = (n + 3) // 4
    i4 = np.arange(n4, dtype=np.uint64)
    ones = np.ones(n4, dtype=np.uint64)
    r = philox4x32_10(i4 & MASK, i4 >> np.uint64(32), ones * np.uint64(stream_id & 0xFFFFFFFF),
                      ones * np.uint64((stream_id >> 32) & 0xFFFFFFFF), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u = [((x >> np.uint64(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0) for x in r]
    out = np.empty((n4, 4), dtype=np.float32)
    for p in range(2):
        rad = np.sqrt(np.float32(-2.0) * np.log(u[2 * p]), dtype=np.float32)
        th = np.float32(6.283185307179586) * u[2 * p + 1]
        out[:, 2 * p] = rad * np.cos(th, dtype=np.float32)
        out[:, 2 * p + 1] = rad * np.sin(th, dtype=np.float32)
    return out.reshape(-1)[:n]
