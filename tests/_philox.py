"""numpy restatement of the draw qlinear_top_p_sample makes (include/qlinear_hip.h): Philox4x32-10 (Salmon, Moraes, Dror, Shaw:
"Parallel random numbers: as easy as 1, 2, 3", SC'11), counter (ctr_lo, ctr_hi, row, 0), key (seed_lo, seed_hi); the uniform
number is the first output word's top 24 bits / 2^24.  Known-answer vectors: the paper's Random123 kat_vectors."""
import numpy as np

M0, M1 = 0xD2511F53, 0xCD9E8D57
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


def philox4x32_10(counter, key):
    c0, c1, c2, c3 = (int(c) & MASK for c in counter)
    k0, k1 = (int(k) & MASK for k in key)
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ c3 ^ k1) & MASK, p0 & MASK
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return c0, c1, c2, c3


def uniform(seed: int, ctr: int, row: int) -> np.float32:
    r = philox4x32_10((ctr & MASK, (ctr >> 32) & MASK, row, 0), (seed & MASK, (seed >> 32) & MASK))[0]
    return np.float32(r >> 8) * np.float32(2.0 ** -24)
