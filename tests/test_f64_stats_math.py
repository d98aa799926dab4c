"""The arithmetic of f64_stats_kernel (cond_kernels.hip), restated in numpy and pinned against numpy's own calls on CPU:
order-preserving 64-bit keys, most-significant-byte-first selection of several ranks at once, and np.mean's summation tree
(8192-element chunks, blocks of 128 with eight running sums, recursive halving) evaluated the way the kernel does it -- one block
per lane and a butterfly for a full chunk, the recursion's uneven cuts for the partial one.  The GPU test
(test_gpu_detect.py::test_float64_order_statistics_equal_numpy) compares the kernel itself with the same numpy calls.
The summation tree is numpy's implementation detail (pairwise sum: 128-element blocks, eight accumulators, an 8192-element
buffer of the reduction loop): it has been this way since numpy 1.9 and is what numpy 2.2.6 (this image) does; a numpy that
changes it would fail THESE tests first, on CPU -- the bit-equality claim of the float64 path is a claim about that arithmetic."""
import numpy as np
import pytest


def f64_key(x):
    x = np.where(x == 0.0, 0.0, x)                      # -0.0 -> +0.0
    b = x.view(np.uint64)
    neg = (b >> np.uint64(63)).astype(bool)
    return np.where(neg, ~b, b | np.uint64(1 << 63))


def f64_from_key(k):
    k = np.uint64(k)
    b = (k & np.uint64((1 << 63) - 1)) if (k >> np.uint64(63)) else ~k
    return np.array([b], np.uint64).view(np.float64)[0]


def radix_select(keys, ranks):
    """Eight passes, one byte each from the top; every rank keeps its prefix and its rank inside the bucket."""
    prefix = [np.uint64(0)] * len(ranks); rem = list(ranks)
    for p in range(8):
        shift = np.uint64(56 - 8 * p)
        for j in range(len(ranks)):
            match = keys if p == 0 else keys[(keys >> (shift + np.uint64(8))) == (prefix[j] >> (shift + np.uint64(8)))]
            hist = np.bincount(((match >> shift) & np.uint64(255)).astype(np.int64), minlength=256)
            b = 0
            while b < 255 and rem[j] >= hist[b]:
                rem[j] -= hist[b]; b += 1
            prefix[j] = prefix[j] | (np.uint64(b) << shift)
    return [f64_from_key(k) for k in prefix]


def block_sum(x):
    """numpy's pairwise sum over one block of at most 128 elements."""
    n = len(x)
    if n < 8:
        r = np.float64(0.0)
        for v in x:
            r = r + v
        return r
    r = [np.float64(v) for v in x[:8]]
    i = 8
    while i < n - (n % 8):
        for j in range(8):
            r[j] = r[j] + x[i + j]
        i += 8
    res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
    for v in x[i:]:
        res = res + v
    return res


def chunk_sum(x):
    """One reduction chunk (<= 8192 elements) the way the kernel evaluates it."""
    n = len(x)
    if n == 8192:
        r = [block_sum(x[128 * l:128 * (l + 1)]) for l in range(64)]            # one block per lane
        o = 1
        while o < 64:                                                           # butterfly: lane l adds lane l ^ o
            r = [r[l] + r[l ^ o] for l in range(64)]
            o <<= 1
        assert all(v == r[0] for v in r)
        return r[0]
    leaves, stack = [], [(0, n)]
    while stack:                                                                # blocks left to right
        a, l = stack.pop()
        if l <= 128:
            leaves.append((a, l)); continue
        n2 = l // 2; n2 -= n2 % 8
        stack.append((a + n2, l - n2)); stack.append((a, n2))
    sums = [block_sum(x[a:a + l]) for a, l in leaves]
    it = iter(sums)

    def fold(l):
        if l <= 128:
            return next(it)
        n2 = l // 2; n2 -= n2 % 8
        left = fold(n2)
        return left + fold(l - n2)
    return fold(n)


def mean_like_the_kernel(x):
    res = np.float64(0.0)
    for i in range(0, len(x), 8192):
        res = res + chunk_sum(x[i:i + 8192])
    return res / np.float64(len(x))


@pytest.mark.parametrize("n", [1, 5, 8, 9, 127, 128, 129, 1000, 8191, 8192, 8193, 16384 + 77, 30011])
def test_summation_tree_equals_numpy_mean(n):
    rng = np.random.default_rng(n)
    x = np.abs(rng.normal(0, 12, n))
    assert mean_like_the_kernel(x) == np.mean(x)


def test_keys_order_like_the_values():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.normal(0, 1e3, 5000), rng.normal(0, 1e-300, 100), [0.0, -0.0, 1e308, -1e308, 5e-324, -5e-324, np.inf, -np.inf]])
    k = f64_key(x)
    order = np.argsort(k, kind="stable")
    assert np.all(np.diff(x[order]) >= 0)
    assert f64_key(np.array([0.0]))[0] == f64_key(np.array([-0.0]))[0]
    for v in (0.0, 1.5, -2.25, 1e-310, -1e308, np.inf):
        assert f64_from_key(f64_key(np.array([v]))[0]) == v


@pytest.mark.parametrize("kind", ["normal", "ties", "two_values", "negative"])
def test_selection_gives_the_order_statistics(kind):
    rng = np.random.default_rng(3)
    n = 20011
    if kind == "normal":
        x = rng.normal(90, 12, n)
    elif kind == "ties":
        x = np.round(rng.normal(90, 12, n) * 4) / 4
    elif kind == "two_values":
        x = np.where(np.arange(n) % 3 == 0, 1.0, 2.0)
    else:
        x = -np.abs(rng.normal(500, 100, n)); x[::7] = 0.0; x[3::7] = -0.0
    ranks = [0, 1, (n - 1) // 2, n // 2, int(0.01 * (n - 1)), int(0.99 * (n - 1)) + 1, n - 1]
    got = radix_select(f64_key(x), ranks)
    want = np.sort(x)[ranks]
    assert np.array_equal(np.asarray(got), want)
