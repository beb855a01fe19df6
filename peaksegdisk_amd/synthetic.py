"""Seeded synthetic coverage for the benchmark configurations.

Piecewise-constant Poisson coverage: background and peak segments alternate, segment lengths
are geometric (mean 2000 bins background / 200 bins peak), rates are U(0.2,1.0) background /
U(4,20) peak, counts ~ Poisson(rate), bin widths ~ U{1..49} bases, bins contiguous from 0.

The random stream is a counter-based splitmix64 implemented here (not numpy's generator), so
a (seed, n_bins) pair names the same data on every machine; tests pin a sha256.
"""
import hashlib

import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(seed, idx):
    """splitmix64 output for state seed + (idx+1)*golden; idx: uint64 array."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (np.asarray(idx, dtype=np.uint64) + np.uint64(1)) * _GOLD
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(seed, stream, n, offset=0):
    """n doubles in (0,1) from stream `stream` of `seed`."""
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        s = np.uint64(seed) ^ (np.uint64(stream) * np.uint64(0xD1B54A32D192ED03))
    u = splitmix64(s, idx) >> np.uint64(11)
    return (u.astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def poisson_from_uniform(lam, u):
    """Poisson counts by CDF inversion (lam <= ~30 here)."""
    lam = np.asarray(lam, dtype=np.float64)
    p = np.exp(-lam)
    cdf = p.copy()
    count = np.zeros(lam.shape, dtype=np.int32)
    active = u > cdf
    k = 0
    while active.any() and k < 400:
        k += 1
        p = p * lam / k
        cdf = cdf + p
        count[active] += 1
        active = active & (u > cdf)
    return count


def poisson_coverage(n_bins, seed=1):
    """Returns (chromStart, chromEnd, count) int32 arrays of length n_bins."""
    n_bins = int(n_bins)
    # enough alternating segments to cover n_bins (mean pair length 2200)
    n_seg = max(16, int(n_bins / 1100.0 * 1.5) + 16)
    while True:
        u_len = uniform01(seed, 1, n_seg)
        u_rate = uniform01(seed, 2, n_seg)
        is_peak = (np.arange(n_seg) % 2) == 1
        mean_len = np.where(is_peak, 200.0, 2000.0)
        q = 1.0 / mean_len
        length = 1 + np.floor(np.log(u_len) / np.log1p(-q)).astype(np.int64)
        if int(length.sum()) >= n_bins:
            break
        n_seg *= 2
    rate = np.where(is_peak, 4.0 + 16.0 * u_rate, 0.2 + 0.8 * u_rate)
    seg_of_bin = np.repeat(np.arange(n_seg), length)[:n_bins]
    lam = rate[seg_of_bin]
    count = poisson_from_uniform(lam, uniform01(seed, 3, n_bins))
    width = 1 + np.floor(uniform01(seed, 4, n_bins) * 49.0).astype(np.int64)
    chrom_end = np.cumsum(width)
    chrom_start = chrom_end - width
    if chrom_end[-1] >= 2 ** 31:
        raise ValueError("synthetic contig exceeds int32 coordinates")
    return chrom_start.astype(np.int32), chrom_end.astype(np.int32), count.astype(np.int32)


def increasing_coverage(n_bins):
    """Worst case of vignettes/Worst_case.Rmd:26-28: count = 1..N, width 1."""
    n_bins = int(n_bins)
    end = np.arange(1, n_bins + 1, dtype=np.int32)
    return (end - 1).astype(np.int32), end, end.copy()


def penalty_grid(n=64, lo=-1.0, hi=5.0):
    """n log-spaced penalties 10^lo..10^hi as strings with 15 significant digits, the way
    R's paste() formats a double."""
    return ["%.15g" % (10.0 ** e) for e in np.linspace(lo, hi, n)]


def write_bedgraph(path, chrom_start, chrom_end, count, chrom="chrSynth"):
    with open(path, "w") as f:
        f.write("".join("%s\t%d\t%d\t%d\n" % (chrom, s, e, c)
                        for s, e, c in zip(chrom_start.tolist(), chrom_end.tolist(),
                                           count.tolist())))


def sha256_of(chrom_start, chrom_end, count):
    h = hashlib.sha256()
    for a in (chrom_start, chrom_end, count):
        h.update(np.ascontiguousarray(a, dtype=np.int32).tobytes())
    return h.hexdigest()
