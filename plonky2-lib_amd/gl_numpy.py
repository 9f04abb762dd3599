"""Vectorised Goldilocks arithmetic on numpy uint64 arrays (host-side helper for building synthetic
circuits and witnesses; not on the prove() path)."""
import numpy as np

P = 0xFFFFFFFF00000001
_P = np.uint64(P)
_EPS = np.uint64(0xFFFFFFFF)
_M32 = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)


def _u(x):
    return np.asarray(x, dtype=np.uint64)


def canon(a):
    a = _u(a)
    return np.where(a >= _P, a - _P, a)


def add(a, b):
    a, b = _u(a), _u(b)
    with np.errstate(over="ignore"):
        s = a + b
        return np.where((s < a) | (s >= _P), s - _P, s)


def sub(a, b):
    a, b = _u(a), _u(b)
    with np.errstate(over="ignore"):
        return np.where(a >= b, a - b, a + (_P - b))


def neg(a):
    a = _u(a)
    return np.where(a == 0, a, _P - a)


def _reduce128(lo, hi):
    with np.errstate(over="ignore"):
        hh, hl = hi >> _S32, hi & _M32
        t0 = lo - hh
        t0 = np.where(lo < hh, t0 - _EPS, t0)
        t1 = (hl << _S32) - hl
        t2 = t0 + t1
        t2 = np.where(t2 < t0, t2 + _EPS, t2)
        return np.where(t2 >= _P, t2 - _P, t2)


_CHUNK = 1 << 16


def mul(a, b):
    """Elementwise product.  Large inputs are processed in cache-sized chunks on a thread pool (numpy
    releases the GIL), which is what makes 2^20-row circuit generation take seconds, not minutes."""
    a, b = np.broadcast_arrays(_u(a), _u(b))
    if a.size > 4 * _CHUNK:
        import os
        from concurrent.futures import ThreadPoolExecutor
        fa, fb = np.ascontiguousarray(a).reshape(-1), np.ascontiguousarray(b).reshape(-1)
        out = np.empty(fa.size, dtype=np.uint64)

        def work(lo):
            out[lo:lo + _CHUNK] = _mul_small(fa[lo:lo + _CHUNK], fb[lo:lo + _CHUNK])
        with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
            list(ex.map(work, range(0, fa.size, _CHUNK)))
        return out.reshape(a.shape)
    return _mul_small(a, b)


def _mul_small(a, b):
    with np.errstate(over="ignore"):
        a0, a1, b0, b1 = a & _M32, a >> _S32, b & _M32, b >> _S32
        p00, p01, p10, p11 = a0 * b0, a0 * b1, a1 * b0, a1 * b1
        mid = p01 + p10
        c1 = (mid < p01).astype(np.uint64)
        lo = p00 + (mid << _S32)
        c2 = (lo < p00).astype(np.uint64)
        hi = p11 + (mid >> _S32) + (c1 << _S32) + c2
    return _reduce128(lo, hi)


def pow_scalar(b, e):
    return pow(int(b), int(e), P)


def inv_scalar(a):
    return pow(int(a), P - 2, P)


def powers(base, n):
    """[1, base, base^2, ...] of length n by repeated doubling."""
    out = np.ones(n, dtype=np.uint64)
    if n > 1:
        out[1] = np.uint64(base % P)
    k = 2
    while k < n:
        m = min(k, n - k)
        out[k:k + m] = mul(out[:m], np.uint64(pow(int(base), k, P)))
        k *= 2
    return out


def root_of_unity(lg):
    return pow(1753635133440165772, 1 << (32 - lg), P)


def rand(rng, shape):
    x = rng.integers(0, 1 << 64, size=shape, dtype=np.uint64)
    return np.where(x >= _P, x - _P, x)
