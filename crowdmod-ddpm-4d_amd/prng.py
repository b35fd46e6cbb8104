"""Counter-based PRNG defined purely in integer arithmetic.

Why it exists: parity fixtures are generated in the build container (where the
reference can be imported) and consumed on the GPU box (where it cannot).  Both
sides must be able to regenerate *bit-identical* fp32 weights, pasts, x_T and
per-step noise z_t without torch and without relying on libm.  Everything here
is uint64 integer math followed by one exactly-rounded int->float conversion.

Streams are addressed by (seed, stream-name, element index): independent of
call order, batch sharding or world size -- sample `i` of a batch gets the same
x_T / z_t on 1 GPU and on 8.

The reference itself draws from torch's global RNG (`torch.randn`,
models/diffusion/ddpm.py:27,211); its stream cannot be reproduced on a device,
so parity runs inject noise generated here on both sides instead.
"""
from __future__ import annotations

import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_C1 = np.uint64(0xBF58476D1CE4E5B9)
_C2 = np.uint64(0x94D049BB133111EB)


def _mix64(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _C1
        z = (z ^ (z >> np.uint64(27))) * _C2
        z = z ^ (z >> np.uint64(31))
    return z


def stream_key(seed: int, name: str) -> np.uint64:
    """64-bit key of a named stream: mixes the seed with a CRC of the name."""
    h = np.uint64(zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        k = _mix64(np.array([np.uint64(seed & 0xFFFFFFFFFFFFFFFF) * _GOLDEN + h], dtype=np.uint64))
    return k[0]


def _raw(key: np.uint64, idx: np.ndarray, lane: int) -> np.ndarray:
    """uint64 hash of (key, element index, lane)."""
    with np.errstate(over="ignore"):
        ctr = idx.astype(np.uint64) * np.uint64(4) + np.uint64(lane)
        return _mix64(key + (ctr + np.uint64(1)) * _GOLDEN)


def uniform_pm1(seed: int, name: str, n: int, offset: int = 0) -> np.ndarray:
    """n fp32 values uniform on [-1, 1): a 24-bit integer scaled exactly."""
    key = stream_key(seed, name)
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    r = _raw(key, idx, 0) >> np.uint64(40)  # 24 bits
    return ((r.astype(np.int64) - (1 << 23)).astype(np.float64) / float(1 << 23)).astype(np.float32)


def normal(seed: int, name: str, n: int, offset: int = 0) -> np.ndarray:
    """n fp32 approximately-N(0,1) values (Irwin-Hall: sum of 12 uniforms - 6).

    The sum of twelve 16-bit uniforms is an exact integer; the only rounding is
    the final float64->float32 cast, so the result is bit-reproducible
    everywhere.  Tails are bounded at +-6, which is irrelevant for parity and
    timing inputs.
    """
    key = stream_key(seed, name)
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    acc = np.zeros(n, dtype=np.int64)
    mask = np.uint64(0xFFFF)
    for lane in range(3):
        r = _raw(key, idx, lane)
        for sh in (0, 16, 32, 48):
            acc += ((r >> np.uint64(sh)) & mask).astype(np.int64)
    # 12 uniforms on {0..65535}/65536, each with mean (65535/2)/65536
    val = (acc.astype(np.float64) - 6.0 * 65535.0) / 65536.0
    return val.astype(np.float32)


def normal_per_sample(seed: int, name: str, sample_ids, per_sample: int, step: int = 0) -> np.ndarray:
    """[len(sample_ids), per_sample] normals addressed by GLOBAL sample id and step.

    Element (i, j) of step s is stream element ((s * 2**20 + id_i) * per_sample + j):
    a batch shard regenerates exactly the rows of the unsharded batch.
    """
    sample_ids = np.asarray(sample_ids, dtype=np.int64)
    out = np.empty((len(sample_ids), per_sample), dtype=np.float32)
    for r, sid in enumerate(sample_ids):
        base = (int(step) * (1 << 20) + int(sid)) * per_sample
        out[r] = normal(seed, name, per_sample, offset=base)
    return out
