"""Numeric utilities with the reference's names and semantics (lnasr/utils.py:100-204).

`lse`, `lse2` and `split_frames` run on the GPU; `create_hamming` is a 400-entry model
constant evaluated on the host; `read_pcm` is file I/O.
"""
import numpy as np

from . import _ffi


def read_pcm(filename: str) -> np.ndarray:
    """Raw PCM: mono, 16 bit, little endian (utils.py:100-104)."""
    with open(filename, "rb") as fp:
        return np.frombuffer(fp.read(), dtype=np.int16)


def split_frames(signal: np.ndarray, L, S):
    """Frames [Nf, L] with a zero-padded tail, Nf = ceil(|len-(L-S)|/S) (utils.py:106-130)."""
    return _ffi.default_ctx().split_frames(np.asarray(signal), int(L), int(S))


def create_hamming(N: int) -> np.ndarray:
    """Symmetric Hamming window 0.54 - 0.46 cos(2 pi n/(N-1)) (utils.py:132-138)."""
    return 0.54 - 0.46 * np.cos(2 * np.pi * np.arange(N) / (N - 1))


def _lse_axis(seq, axis, base2):
    seq = np.asarray(seq)
    nd = seq.ndim
    if axis < 0:
        axis += nd
    n = seq.shape[axis]
    if n == 0:
        # the reference drops EVERY zero-length dimension here (utils.py:167-168)
        return np.full(tuple(d for d in seq.shape if d != 0), -np.inf)
    if n == 1:
        return seq.take(0, axis)
    outer = int(np.prod(seq.shape[:axis], dtype=np.int64))
    inner = int(np.prod(seq.shape[axis + 1:], dtype=np.int64))
    out_shape = seq.shape[:axis] + seq.shape[axis + 1:]
    res = _ffi.default_ctx().lse3(np.asarray(seq, dtype=np.float64).reshape(outer, n, inner), base2)
    return res.reshape(out_shape) if out_shape else np.float64(res.reshape(()))


def _lse(seq, axis, base2):
    seq = np.asarray(seq)
    if axis is None:                      # utils.py:157-158
        return _lse_axis(seq.ravel(), 0, base2)
    if isinstance(axis, tuple):           # utils.py:159-165: last listed axis first
        out = seq
        for ax in reversed(axis):
            out = _lse_axis(out, ax, base2)
        return out
    return _lse_axis(seq, int(axis), base2)


def lse(seq: np.ndarray, axis=None) -> np.ndarray:
    """log(sum(exp(seq))) over `axis` (None: all elements; tuple: ascending axes) (utils.py:140-179)."""
    return _lse(seq, axis, False)


def lse2(seq: np.ndarray, axis=None) -> np.ndarray:
    """Base-2 log-sum-exp (utils.py:181-204)."""
    return _lse(seq, axis, True)
