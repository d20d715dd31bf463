"""cluster id of the synthetic clustered generator rows (misc.hip ph_synth_clustered_kernel)"""
import numpy as np
M64 = (1 << 64) - 1


def mix64(x):
    x ^= x >> 30
    x = (x * 0xBF58476D1CE4E5B9) & M64
    x ^= x >> 27
    x = (x * 0x94D049BB133111EB) & M64
    x ^= x >> 31
    return x


def cluster_of(first, count, seed=42, ncl=1000):
    out = np.empty(count, dtype=np.int64)
    for r in range(count):
        key = (seed + first + r) & M64
        out[r] = (mix64((key * 0xA24BAED4963EE407 + 0x9FB21C651E98DF25) & M64) * ncl) >> 64
    return out


