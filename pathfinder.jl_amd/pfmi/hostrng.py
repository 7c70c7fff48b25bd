"""Host-side counter-based RNG (Philox4x32-10, NumPy) for the seed hierarchy and synthetic inputs.

The reference seeds its runs with ``rand!(rng, UInt64[nruns])`` (src/multipath.jl:162) and its fits
with ``rand!(rng, UInt64[L])`` (src/elbo.jl:2); Julia's Xoshiro stream cannot be reproduced outside
Julia, so this module provides the same *structure* (master rng -> run seeds -> fit seeds) on the
generator the device kernels use.  Bit-identical to pf_philox4x32_10 in csrc/pfmi_common.h.
"""
import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr: (..., 4) uint32, key: (k0, k1) ints or arrays broadcastable to ctr[..., 0] -> (..., 4) uint32"""
    c = np.asarray(ctr, dtype=np.uint64) & _MASK
    c0, c1, c2, c3 = c[..., 0], c[..., 1], c[..., 2], c[..., 3]
    k0 = np.asarray(key[0], dtype=np.uint64) & _MASK
    k1 = np.asarray(key[1], dtype=np.uint64) & _MASK
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & _MASK
        n1 = p1 & _MASK
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & _MASK
        n3 = p0 & _MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(_W0)) & _MASK
        k1 = (k1 + np.uint64(_W1)) & _MASK
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def _native():
    """pfmi_host_rand_u64 of libpfmi.so (plain host code) when the library is built; None -> NumPy below."""
    global _NATIVE
    if _NATIVE is None:
        try:
            import ctypes as C
            from . import _lib
            f = _lib.lib().pfmi_host_rand_u64
            f.argtypes = [C.c_uint64, C.c_uint64, C.c_int64, C.c_uint32, C.POINTER(C.c_uint64)]
            _NATIVE = f
        except Exception:
            _NATIVE = False
    return _NATIVE or None


_NATIVE = None


def rand_u64_multi(rngs, counts):
    """[r.rand_u64(n) for r, n in zip(rngs, counts)] in ONE vectorised Philox evaluation (advances every rng)."""
    counts = [int(n) for n in counts]
    tot = sum(counts)
    if tot == 0:
        return [np.zeros(0, dtype=np.uint64) for _ in counts]
    f = _native()
    if f is not None and len(counts) > 1 and min(counts) == max(counts):
        # every generator the same number of values (the runs' starting points, their seed streams): ONE native call
        import ctypes as C
        from . import _lib
        n, m = counts[0], len(counts)
        sd = np.array([r.seed for r in rngs], dtype=np.uint64)
        t0 = np.array([r.counter for r in rngs], dtype=np.uint64)
        out = np.empty((m, n), dtype=np.uint64)
        u64p = C.POINTER(C.c_uint64)
        rc = _lib.lib().pfmi_host_rand_u64_multi(C.c_int32(m), sd.ctypes.data_as(u64p), t0.ctypes.data_as(u64p), C.c_int64(n),
                                                 C.c_uint32(HostRNG.STREAM), out.ctypes.data_as(u64p))
        if rc == 0:
            for r in rngs:
                r.counter += n
            return list(out)
    if f is not None:
        import ctypes as C
        res = []
        for r, n in zip(rngs, counts):
            out = np.empty(n, dtype=np.uint64)
            f(C.c_uint64(r.seed), C.c_uint64(r.counter), C.c_int64(n), C.c_uint32(HostRNG.STREAM), out.ctypes.data_as(C.POINTER(C.c_uint64)))
            r.counter += n
            res.append(out)
        return res
    t = np.concatenate([np.arange(r.counter, r.counter + n, dtype=np.uint64) for r, n in zip(rngs, counts)])
    sd = np.concatenate([np.full(n, r.seed, dtype=np.uint64) for r, n in zip(rngs, counts)])
    ctr = np.stack([t & _MASK, t >> np.uint64(32), np.full_like(t, HostRNG.STREAM), np.zeros_like(t)], axis=-1)
    x = philox4x32_10(ctr, (sd & _MASK, sd >> np.uint64(32))).astype(np.uint64)
    out = x[..., 0] | (x[..., 1] << np.uint64(32))
    res, o = [], 0
    for r, n in zip(rngs, counts):
        res.append(out[o:o + n]); o += n
        r.counter += n
    return res


def rand_u64(seed, t, stream):
    """64 random bits for counters t (array) of `seed` on `stream` (== pf_rand_u64)."""
    t = np.atleast_1d(np.asarray(t, dtype=np.uint64))
    f = _native()
    if f is not None and t.ndim == 1 and len(t) > 0 and np.array_equal(t, np.arange(t[0], t[0] + np.uint64(len(t)), dtype=np.uint64)):
        import ctypes as C
        out = np.empty(len(t), dtype=np.uint64)
        f(C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), C.c_uint64(int(t[0])), C.c_int64(len(t)), C.c_uint32(int(stream)),
          out.ctypes.data_as(C.POINTER(C.c_uint64)))
        return out
    ctr = np.stack([t & _MASK, t >> np.uint64(32), np.full_like(t, stream), np.zeros_like(t)], axis=-1)
    seed = int(seed)
    x = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)).astype(np.uint64)
    return x[..., 0] | (x[..., 1] << np.uint64(32))


class HostRNG:
    """Minimal AbstractRNG stand-in: seedable, copyable, counter-based."""
    STREAM = 7   # host streams never collide with device streams 0 (normals), 1 (resample), 2 (norep)

    def __init__(self, seed=0):
        self.seed_(seed)

    def seed_(self, seed):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.counter = 0
        return self

    def copy(self):
        r = HostRNG(self.seed)
        r.counter = self.counter
        return r

    def rand_u64(self, n):
        out = rand_u64(self.seed, np.arange(self.counter, self.counter + n, dtype=np.uint64), self.STREAM)
        self.counter += n
        return out

    def rand(self, n):
        """uniform [0,1) doubles (53 bits)"""
        return (self.rand_u64(n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)

    def randn(self, n):
        m = (n + 1) // 2
        u1 = ((self.rand_u64(m) >> np.uint64(11)).astype(np.float64) + 0.5) / 9007199254740992.0
        u2 = self.rand(m)
        r = np.sqrt(-2.0 * np.log(u1))
        z = np.concatenate([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)])
        return z[:n]
