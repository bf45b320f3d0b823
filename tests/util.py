import numpy as np


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.int32) if a.dtype == np.float32 else a


def same_bits(a, b):
    return np.array_equal(bits(a), bits(b))


def gauss(n, d, seed):
    return np.random.default_rng(seed).standard_normal((n, d), dtype=np.float32)


def recall_at_k(found, truth):
    hit = 0
    for f, t in zip(found, truth):
        hit += len(set(int(x) for x in f if x >= 0) & set(int(x) for x in t))
    return hit / float(truth.shape[0] * truth.shape[1])
