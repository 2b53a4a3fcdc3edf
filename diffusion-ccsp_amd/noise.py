"""Counter-based noise stream shared by every implementation of the sampling path.

The reference draws its Gaussian noise with ``torch.randn(shape, device=...)`` in a fixed call
order (networks/ddpm.py:273 init, :255 ancestral, :292/:963 ULA, :1020/:1037 MALA).  "Identical
seeds" across a CPU reference and a HIP sampler cannot mean torch's device generators, so the
build owns one stateless generator and every side regenerates the same stream from it:

    value(call c, global row n, column p) = BoxMuller(Philox4x32-10(key=seed,
                                              counter=(n, c, p >> 2, stream)))[p & 3]

* ``stream`` 0 = normal draws ``z`` (``randn(N, P)``), 1 = uniform draws ``u`` (``rand(N)``).
* rows are *global* node indices, so a rank that owns rows [r0, r1) of a sharded batch
  regenerates exactly its slice (parity-mode noise).
* the uniforms fed to Box-Muller are 24-bit (exact in fp32): u1 = ((r>>8)+1) * 2^-24 in (0, 1],
  u2 = (r>>8) * 2^-24 in [0, 1).  Host code (this file, oracle/ccsp_oracle.c) evaluates
  log/sin/cos in float64 and rounds to fp32; the HIP kernel evaluates them in fp32, so device
  draws agree with host draws to a few ulp (checked in tests/test_noise.py).

This file is the numpy statement of the generator (used by the golden-vector capture and tests);
csrc/ccsp_philox.h is the device statement.
"""
import numpy as np

PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = np.uint32(0x9E3779B9)
PHILOX_W1 = np.uint32(0xBB67AE85)

STREAM_NORMAL = 0
STREAM_UNIFORM = 1


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """vectorised Philox4x32-10; all inputs uint32 arrays (broadcastable); returns 4 uint32 arrays"""
    c0 = np.asarray(c0, dtype=np.uint32)
    c1 = np.asarray(c1, dtype=np.uint32)
    c2 = np.asarray(c2, dtype=np.uint32)
    c3 = np.asarray(c3, dtype=np.uint32)
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    mask = np.uint64(0xFFFFFFFF)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = PHILOX_M0 * c0.astype(np.uint64)
            p1 = PHILOX_M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & mask).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & mask).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(PHILOX_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(PHILOX_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def _box_muller(ra, rb):
    u1 = ((ra >> np.uint32(8)).astype(np.float64) + 1.0) * (1.0 / 16777216.0)
    u2 = (rb >> np.uint32(8)).astype(np.float64) * (1.0 / 16777216.0)
    rad = np.sqrt(-2.0 * np.log(u1))
    ang = 2.0 * np.pi * u2
    return rad * np.cos(ang), rad * np.sin(ang)


def normal(seed, call, n_rows, n_cols, row0=0):
    """the fp32 ``randn(n_rows, n_cols)`` of call index ``call`` for global rows row0..
    ``call`` may be an int (-> [n_rows, n_cols]) or a 1-D array (-> [len(call), n_rows, n_cols])"""
    seed = int(seed)
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    scalar = np.isscalar(call)
    calls = np.atleast_1d(np.asarray(call, dtype=np.uint32))[:, None, None]
    nsub = (n_cols + 3) // 4
    rows = (np.arange(n_rows, dtype=np.uint64) + np.uint64(row0)).astype(np.uint32)[None, :, None]
    sub = np.arange(nsub, dtype=np.uint32)[None, None, :]
    r0, r1, r2, r3 = philox4x32_10(rows, calls, sub, np.uint32(STREAM_NORMAL), k0, k1)
    z0, z1 = _box_muller(r0, r1)
    z2, z3 = _box_muller(r2, r3)
    z = np.stack([z0, z1, z2, z3], axis=-1).reshape(calls.shape[0], n_rows, nsub * 4)
    z = np.ascontiguousarray(z[:, :, :n_cols]).astype(np.float32)
    return z[0] if scalar else z


def uniform(seed, call, n_rows, row0=0):
    """the fp32 ``rand(n_rows)`` of uniform-call index ``call`` (values in [0, 1))"""
    seed = int(seed)
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    scalar = np.isscalar(call)
    calls = np.atleast_1d(np.asarray(call, dtype=np.uint32))[:, None]
    rows = (np.arange(n_rows, dtype=np.uint64) + np.uint64(row0)).astype(np.uint32)[None, :]
    r0, _, _, _ = philox4x32_10(rows, calls, np.uint32(0), np.uint32(STREAM_UNIFORM), k0, k1)
    u = ((r0 >> np.uint32(8)).astype(np.float64) * (1.0 / 16777216.0)).astype(np.float32)
    return u[0] if scalar else u


def n_normal_calls(T, samples_per_step):
    """number of randn(N,P) calls of one chain: init + T ancestral + sum_t S_t"""
    if np.isscalar(samples_per_step):
        return 1 + T * (1 + int(samples_per_step))
    return 1 + T + int(np.sum(samples_per_step))


def normal_stream(seed, n_calls, n_rows, n_cols, row0=0, chunk=512):
    """[n_calls, n_rows, n_cols] fp32 -- the INJECTED-mode tensor equal to the PHILOX-mode stream"""
    out = np.empty((n_calls, n_rows, n_cols), dtype=np.float32)
    for c in range(0, n_calls, chunk):
        e = min(n_calls, c + chunk)
        out[c:e] = normal(seed, np.arange(c, e), n_rows, n_cols, row0)
    return out


def uniform_stream(seed, n_calls, n_rows, row0=0):
    if n_calls == 0:
        return np.empty((0, n_rows), dtype=np.float32)
    return uniform(seed, np.arange(n_calls), n_rows, row0)
