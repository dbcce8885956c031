"""ctypes binding of liblnasr_hip.so (include/lnasr_hip.h) -- the only caller of the C-ABI.

There is NO CPU fallback: if the HIP library is missing or no gfx950 device is visible,
every compute entry point raises.  Build the library with `python ln-asr_amd/build.py`.
"""
import ctypes as C
import os
import threading

import numpy as np

F64, F32 = 0, 1
_HERE = os.path.dirname(os.path.abspath(__file__))
# LNASR_LIB selects another build of the same library (the host-sanitizer build of build.py --host-asan)
_LIB_PATH = os.environ.get("LNASR_LIB") or os.path.join(_HERE, "liblnasr_hip.so")
UNIQUE_ID_BYTES = 128


class LnasrError(RuntimeError):
    pass


class MfccCfgStruct(C.Structure):
    _fields_ = [("fs", C.c_int), ("frame_len", C.c_int), ("frame_step", C.c_int), ("fft_n", C.c_int),
                ("n_mel", C.c_int), ("n_cep", C.c_int), ("hpf_alpha", C.c_double)]


_vp, _i, _sz, _dbl, _i64 = C.c_void_p, C.c_int, C.c_size_t, C.c_double, C.c_int64
_P = C.POINTER

# name -> argtypes (every function returns int unless listed in _RESTYPES)
SIGNATURES = {
    "lnasr_abi_version": [],
    "lnasr_last_error": [],
    "lnasr_device_count": [_P(_i)],
    "lnasr_ctx_create": [_i, _P(_vp)],
    "lnasr_ctx_destroy": [_vp],
    "lnasr_sync": [_vp],
    "lnasr_device_sync": [_vp],
    "lnasr_reload_switches": [],
    "lnasr_host_alloc": [_vp, _sz, _P(_vp)],
    "lnasr_host_free": [_vp, _vp],
    "lnasr_device_info": [_vp, C.c_char_p, _i, _P(_i), _P(_sz)],
    "lnasr_hash128": [_vp, _sz, C.c_uint64, _P(C.c_uint64)],
    "lnasr_hash128_begin": [_P(_vp), _P(_sz), _i, C.c_uint64, _P(_vp)],
    "lnasr_hash128_end": [_vp, _P(C.c_uint64)],
    "lnasr_debug_require_device_ptr": [_vp],
    "lnasr_debug_fallback_columns": [_vp, _P(C.c_uint)],
    "lnasr_debug_clock": [_vp, _P(_dbl)],
    "lnasr_dev_alloc": [_vp, _sz, _P(_vp)],
    "lnasr_dev_free": [_vp, _vp],
    "lnasr_memcpy_h2d": [_vp, _vp, _vp, _sz],
    "lnasr_memcpy_d2h": [_vp, _vp, _vp, _sz],
    "lnasr_memset": [_vp, _vp, _i, _sz],
    "lnasr_prof_enable": [_vp, _i],
    "lnasr_prof_reset": [_vp],
    "lnasr_prof_count": [_vp, _P(_i)],
    "lnasr_prof_get": [_vp, _i, C.c_char_p, _i, _P(_i64), _P(_dbl)],
    "lnasr_model_upload": [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i],
    "lnasr_model_upload_full": [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "lnasr_hmm_upload": [_vp, _i, _vp, _vp],
    "lnasr_model_is_diag": [_vp, _P(_i)],
    "lnasr_model_conditioning": [_vp, _P(_dbl), _P(_i)],
    "lnasr_model_range": [_vp, _P(_dbl), _P(_i)],
    "lnasr_discrete_map_b": [_vp, _vp, _i, _i, _vp, _i, _i, _i, _vp],
    "lnasr_gmm_logprob": [_vp, _vp, _i, _i, _vp, _i, _vp, _vp],
    "lnasr_gauss1d": [_vp, _vp, _sz, _dbl, _dbl, _i, _vp],
    "lnasr_gauss_multi": [_vp, _vp, _sz, _i, _vp, _vp, _i, _vp],
    "lnasr_gauss_mixture": [_vp, _vp, _i, _vp, _sz, _i, _vp, _vp, _i, _vp],
    "lnasr_viterbi": [_vp, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp],
    "lnasr_decode": [_vp, _vp, _i, _i, _vp, _vp, _vp],
    "lnasr_forward": [_vp, _vp, _i, _i, _vp, _i, _vp, _vp],
    "lnasr_backward": [_vp, _vp, _i, _i, _vp, _i, _vp],
    "lnasr_calc_prob": [_vp, _vp, _i, _i, _vp, _vp],
    "lnasr_calc_prob_p": [_vp, _vp, _i, _i, _vp, _i, _vp],
    "lnasr_baumwelch": [_vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _vp],
    "lnasr_estimate_discrete": [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "lnasr_estimate_gmm": [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _dbl, _vp, _vp, _vp, _vp, _vp],
    "lnasr_bw_stats_len": [_vp, _P(_sz)],
    "lnasr_bw_accumulate": [_vp, _vp, _i, _i, _vp, _vp],
    "lnasr_bw_accumulate_p": [_vp, _vp, _i, _i, _vp, _i, _vp],
    "lnasr_bw_mstep": [_vp, _vp, _dbl, _vp, _vp, _vp, _vp, _vp],
    "lnasr_lse": [_vp, _vp, _sz, _sz, _sz, _i, _vp],
    "lnasr_num_frames": [_i64, _i, _i, _P(_i64)],
    "lnasr_preemphasis": [_vp, _vp, _i64, _dbl, _vp],
    "lnasr_split_frames": [_vp, _vp, _i64, _i, _i, _vp],
    "lnasr_mfcc": [_vp, _vp, _i, _vp, _i, _P(MfccCfgStruct), _vp, _vp, _vp, _vp, _vp],
    "lnasr_pad_rows": [_vp, _vp, _vp, _i, _i, _i64, _vp],
    "lnasr_comm_unique_id": [_vp],
    "lnasr_comm_init": [_vp, _vp, _i, _i],
    "lnasr_comm_destroy": [_vp],
    "lnasr_comm_info": [_vp, _P(_i), _P(_i)],
    "lnasr_allgather_u32": [_vp, _vp, _sz, _vp],
    "lnasr_allgather_paths": [_vp, _vp, _sz, _i, _vp],
    "lnasr_allreduce_f64": [_vp, _vp, _sz],
}
_RESTYPES = {"lnasr_last_error": C.c_char_p, "lnasr_ctx_destroy": None}

_lib = None
_lock = threading.Lock()


def lib():
    """Load liblnasr_hip.so (once).  Raises LnasrError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(_LIB_PATH):
                raise LnasrError("%s not found: build it with `python ln-asr_amd/build.py` "
                                 "(there is no CPU fallback)" % _LIB_PATH)
            l = C.CDLL(_LIB_PATH)
            for name, args in SIGNATURES.items():
                fn = getattr(l, name)
                fn.argtypes = args
                fn.restype = _RESTYPES.get(name, C.c_int)
            if l.lnasr_abi_version() != 1:
                raise LnasrError("ABI version mismatch")
            _lib = l
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().lnasr_last_error()
        raise LnasrError("lnasr error %d: %s" % (rc, msg.decode() if msg else "?"))


def np_dtype(code):
    return np.float32 if code == F32 else np.float64


def dtype_code(precision):
    return F32 if np.dtype(precision) == np.float32 else F64


class DeviceArray:
    """A device-resident, C-contiguous array (HBM) owned by a Context."""

    def __init__(self, ctx, shape, dtype):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = C.c_void_p()
        check(lib().lnasr_dev_alloc(ctx.handle, self.nbytes, C.byref(p)))
        self.ptr = p.value

    @classmethod
    def from_host(cls, ctx, arr, dtype=None):
        arr = np.ascontiguousarray(arr, dtype=dtype)
        d = cls(ctx, arr.shape, arr.dtype)
        check(lib().lnasr_memcpy_h2d(ctx.handle, d.ptr, arr.ctypes.data, d.nbytes))
        return d

    def view(self, shape):
        """The same device memory under another shape (same element count); the view keeps its parent alive and
        does not own the allocation.  Lets a batch stay in HBM between entry points, e.g. MFCC features [F, D] ->
        decode input [U, T, D]."""
        shape = tuple(int(s) for s in shape)
        if int(np.prod(shape, dtype=np.int64)) * self.dtype.itemsize != self.nbytes:
            raise ValueError("view of %s as %s changes the size" % (self.shape, shape))
        v = DeviceArray.__new__(DeviceArray)
        v.ctx, v.shape, v.dtype, v.nbytes, v.ptr = self.ctx, shape, self.dtype, self.nbytes, self.ptr
        v._parent = self
        return v

    def to_host(self):
        out = np.empty(self.shape, dtype=self.dtype)
        check(lib().lnasr_memcpy_d2h(self.ctx.handle, out.ctypes.data, self.ptr, self.nbytes))
        return out

    def rows_to_host(self, first, count, width):
        """`count` rows of `width` elements starting at row `first` of the array seen as [rows, width] -> numpy [count, width]
        (e.g. alpha[T-1] of a resident [1,T,N] trellis without moving the other T-1 rows)."""
        total = self.nbytes // (int(width) * self.dtype.itemsize)
        if not (0 <= first and first + count <= total):
            raise ValueError("rows %d..%d of %d" % (first, first + count, total))
        out = np.empty((int(count), int(width)), dtype=self.dtype)
        off = int(first) * int(width) * self.dtype.itemsize
        check(lib().lnasr_memcpy_d2h(self.ctx.handle, out.ctypes.data, self.ptr + off, out.nbytes))
        return out

    def free(self):
        if getattr(self, "_parent", None) is not None:      # a view: the parent owns the memory
            self._parent = None
        elif self.ptr and self.ctx.handle:
            lib().lnasr_dev_free(self.ctx.handle, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _PinnedBlock:
    """Owner of one lnasr_host_alloc block.  `buffer()` hands out a ctypes view that references this object, and a
    numpy array made over that view references the view (`.base`): the block lives exactly as long as any array over
    it and is freed when the last one is collected."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = max(1, int(nbytes))
        p = C.c_void_p()
        check(lib().lnasr_host_alloc(ctx.handle, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def buffer(self):
        buf = (C.c_char * self.nbytes).from_address(self.ptr)
        buf._lnasr_owner = self                # the view keeps the block alive (no cycle: the block holds no view)
        return buf

    def __del__(self):
        try:
            if self.ptr:                       # a block that outlives its (closed) context is still freed: null ctx
                lib().lnasr_host_free(self.ctx.handle, self.ptr)
        except Exception:
            pass
        self.ptr = None


class _FingerprintJob:
    """A parameter fingerprint computed in the background (lnasr_hash128_begin / _end): started before a call's kernels are
    enqueued, collected with result() when the call is about to hand its answer back.  result() equals what
    Context._fingerprint returns for the same arrays."""

    def __init__(self, head, arrays):
        self._arrays = [a if a.flags.c_contiguous else np.ascontiguousarray(a) for a in arrays]   # kept alive until result()
        n = len(self._arrays)
        ptrs = (C.c_void_p * n)(*[a.ctypes.data if a.size else None for a in self._arrays])
        sizes = (C.c_size_t * n)(*[a.nbytes for a in self._arrays])
        job = C.c_void_p()
        check(lib().lnasr_hash128_begin(ptrs, sizes, n, 0, C.byref(job)))
        self._job, self._head = job, head

    def result(self):
        job, self._job = self._job, None
        out = (C.c_uint64 * (2 * len(self._arrays)))()
        check(lib().lnasr_hash128_end(job, out))
        fp = [self._head] + [(out[2 * k], out[2 * k + 1]) for k, a in enumerate(self._arrays) if a.size]
        self._arrays = None
        return tuple(fp)

    def __del__(self):
        try:
            if self._job:
                lib().lnasr_hash128_end(self._job, None)
        except Exception:
            pass


def _ptr(x):
    """void* of a numpy array / DeviceArray / None."""
    if x is None:
        return None
    if isinstance(x, DeviceArray):
        return x.ptr
    return x.ctypes.data


def _in(x, dtype):
    """Host arrays are made C-contiguous of `dtype`; DeviceArrays pass through."""
    if isinstance(x, DeviceArray):
        if x.dtype != np.dtype(dtype):
            raise LnasrError("device array dtype %s, expected %s" % (x.dtype, np.dtype(dtype)))
        return x
    return np.ascontiguousarray(x, dtype=dtype)


class Context:
    """One process = one GPU = one Context (one HIP stream)."""

    def __init__(self, device=None):
        n = C.c_int(0)
        check(lib().lnasr_device_count(C.byref(n)))
        if device is None:
            if "LNASR_DEVICE" in os.environ:
                device = int(os.environ["LNASR_DEVICE"])
            else:
                device = int(os.environ.get("LOCAL_RANK", "0"))
                if n.value == 1:
                    # a launcher that isolates one GPU per rank (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per task)
                    # still sets LOCAL_RANK = k: the rank's only visible device is the right one
                    device = 0
        if n.value > 0 and not 0 <= int(device) < n.value:
            # an explicit request is never wrapped: ranks that silently share a GPU would report success on the
            # wrong device
            raise LnasrError("device %d requested (argument / LNASR_DEVICE / LOCAL_RANK) but only %d visible"
                             % (device, n.value))
        h = C.c_void_p()
        check(lib().lnasr_ctx_create(int(device), C.byref(h)))
        self.handle = h.value
        self.device = int(device)
        self.model_owner = None       # (id(obj), version) of the model currently uploaded
        self.model_shape = None       # (N, M, D) or (N,)
        self._model_fp = None         # fingerprint of the parameters the device holds (see upload_*)

    def close(self):
        if self.handle:
            lib().lnasr_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- runtime -----------------------------------------------------------------------
    def sync(self):
        check(lib().lnasr_sync(self.handle))

    def device_sync(self):
        """hipDeviceSynchronize on this context's device (every stream drained): the bracket of a timed region."""
        check(lib().lnasr_device_sync(self.handle))

    def pinned_empty(self, shape, dtype=np.float64):
        """A numpy array in page-locked host memory (lnasr_host_alloc).  Entry points recognise such buffers and move
        them at PCIe rate without a staging copy; the memory is freed when the last array over it is collected."""
        dtype = np.dtype(dtype)
        shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        n = int(np.prod(shape, dtype=np.int64))
        buf = _PinnedBlock(self, n * dtype.itemsize).buffer()
        return np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)

    def pinned_copy(self, arr):
        out = self.pinned_empty(np.shape(arr), np.asarray(arr).dtype)
        out[...] = arr
        return out

    def device_info(self):
        name = C.create_string_buffer(256)
        cu = C.c_int(0)
        mem = C.c_size_t(0)
        check(lib().lnasr_device_info(self.handle, name, 256, C.byref(cu), C.byref(mem)))
        return {"name": name.value.decode(), "cu_count": cu.value, "hbm_bytes": mem.value}

    def empty(self, shape, dtype=np.float64):
        return DeviceArray(self, shape, dtype)

    def to_device(self, arr, dtype=None):
        return DeviceArray.from_host(self, arr, dtype)

    def fallback_columns(self):
        """Diagnostic (lnasr_debug_fallback_columns): columns the column-owned forward / backward kernels redid in the exact
        log-domain form since the last call; reading clears the counter."""
        n = C.c_uint(0)
        check(lib().lnasr_debug_fallback_columns(self.handle, C.byref(n)))
        return n.value

    def debug_clock(self):
        """Shader clock (GHz) directly behind whatever this context's stream ran last (lnasr_debug_clock)."""
        g = C.c_double(0)
        check(lib().lnasr_debug_clock(self.handle, C.byref(g)))
        return g.value

    def prof_enable(self, on=True):
        check(lib().lnasr_prof_enable(self.handle, 1 if on else 0))

    def prof_reset(self):
        check(lib().lnasr_prof_reset(self.handle))

    def prof_report(self):
        """{kernel name: (launches, total_ms)} from HIP events on this context's stream."""
        n = C.c_int(0)
        check(lib().lnasr_prof_count(self.handle, C.byref(n)))
        out = {}
        for i in range(n.value):
            name = C.create_string_buffer(128)
            cnt = C.c_int64(0)
            ms = C.c_double(0)
            check(lib().lnasr_prof_get(self.handle, i, name, 128, C.byref(cnt), C.byref(ms)))
            if cnt.value:
                out[name.value.decode()] = (cnt.value, ms.value)
        return out

    # -- models --------------------------------------------------------------------------
    # The reference API re-reads self.A / self.mu / ... on every call, so the mirror classes hand the model over on
    # every call too.  Fingerprinting the parameters (lnasr_hash128: ~0.2 ms for the 25 MB of a 128-state
    # full-covariance model) and skipping the upload when the device already holds exactly these bytes is far cheaper
    # than the transfer + LU inversion + coefficient packing it saves, and unlike a dirty flag it also sees in-place
    # edits.  The arrays are fingerprinted AS GIVEN (dtype and strides included), so a float32 or non-contiguous
    # parameter is converted only when it actually has to be uploaded.
    @staticmethod
    def _fp_head(tag, arrays, extra):
        return repr((tag,) + tuple((a.shape, a.dtype.str) for a in arrays) + tuple(extra))

    @staticmethod
    def _fingerprint(tag, arrays, *extra):
        out = (C.c_uint64 * 2)()
        fp = [Context._fp_head(tag, arrays, extra)]
        for k, a in enumerate(arrays):
            if not a.size:
                continue
            if not a.flags.c_contiguous:
                a = np.ascontiguousarray(a)
            check(lib().lnasr_hash128(a.ctypes.data, a.nbytes, k, out))
            fp.append((out[0], out[1]))
        return tuple(fp)

    def gmm_fingerprint_job(self, A, pi, w, mu, sigma, diag=False, force_full=False):
        """The fingerprint upload_gmmhmm would compute for these parameters, started in the background."""
        arrays = tuple(np.asarray(x) for x in (A, pi, w, mu, sigma))
        return _FingerprintJob(self._fp_head("gmm", arrays, (bool(diag), bool(force_full))), arrays)

    def upload_hmm(self, A, pi):
        self.model_owner = None
        A, pi = np.asarray(A), np.asarray(pi)
        N = A.shape[0]
        if A.shape != (N, N) or pi.shape != (N,):
            raise ValueError("A must be [N,N] and pi [N]; got %s, %s" % (A.shape, pi.shape))
        fp = self._fingerprint("hmm", (A, pi))
        if fp == self._model_fp:
            return
        self._model_fp = None
        A = _in(A, np.float64)
        pi = _in(pi, np.float64)
        check(lib().lnasr_hmm_upload(self.handle, N, _ptr(A), _ptr(pi)))
        self.model_shape = (N,)
        self._model_fp = fp

    def upload_hmm_device(self, A, pi, token=None):
        """The same upload from DeviceArrays (the A and pi an on-device M-step left in HBM: lnasr.hmm.HMM.train): device to
        device, no fingerprint -- nobody outside can have edited these arrays; `token` is recorded as what the context holds."""
        for x in (A, pi):
            if not isinstance(x, DeviceArray) or x.ctx is not self or not x.ptr or x.dtype != np.float64:
                raise LnasrError("upload_hmm_device takes float64 DeviceArrays of this context")
        N = A.shape[0]
        if A.shape != (N, N) or pi.shape != (N,):
            raise ValueError("A must be [N,N] and pi [N]; got %s, %s" % (A.shape, pi.shape))
        self.model_owner = None
        self._model_fp = None
        check(lib().lnasr_hmm_upload(self.handle, N, A.ptr, pi.ptr))
        self.model_shape = (N,)
        self._model_fp = token

    def upload_gmmhmm(self, A, pi, w, mu, sigma, diag=False, force_full=False, owner=None):
        """sigma: [N,M,D,D] (diag=False) or [N,M,D] (diag=True).  owner: id of the model object these parameters
        belong to (recorded as self.model_owner: the mirror classes speculate only on a model they uploaded themselves)."""
        self.model_owner = owner
        A, pi, w, mu, sigma = (np.asarray(x) for x in (A, pi, w, mu, sigma))
        N, M, D = mu.shape
        if A.shape != (N, N) or pi.shape != (N,) or w.shape != (N, M):
            raise ValueError("inconsistent GMM-HMM shapes")
        if sigma.shape != ((N, M, D) if diag else (N, M, D, D)):
            raise ValueError("sigma shape %s does not match (N,M,D)=%s" % (sigma.shape, (N, M, D)))
        if force_full and diag:
            raise ValueError("force_full needs a full covariance")
        fp = self._fingerprint("gmm", (A, pi, w, mu, sigma), bool(diag), bool(force_full))
        if fp == self._model_fp:
            return
        self._model_fp = None
        A, pi, w, mu, sigma = (_in(x, np.float64) for x in (A, pi, w, mu, sigma))
        if force_full:
            check(lib().lnasr_model_upload_full(self.handle, N, M, D, _ptr(A), _ptr(pi), _ptr(w), _ptr(mu), _ptr(sigma)))
        else:
            check(lib().lnasr_model_upload(self.handle, N, M, D, _ptr(A), _ptr(pi), _ptr(w), _ptr(mu), _ptr(sigma),
                                           1 if diag else 0))
        self.model_shape = (N, M, D)
        self._model_fp = fp

    def upload_gmmhmm_device(self, A, pi, w, mu, sigma, diag=False, force_full=False, owner=None, token=None):
        """The same upload from DeviceArrays (the parameters an on-device M-step left in HBM: lnasr.gmmhmm.GMMHMM.train):
        device-to-device copies, no fingerprint -- nobody outside can have edited these arrays.  `token` becomes the
        context's record of what it holds; it never equals a fingerprint of host arrays, so a later host upload is
        checked as usual."""
        for x in (A, pi, w, mu, sigma):
            if not isinstance(x, DeviceArray) or x.ctx is not self or not x.ptr or x.dtype != np.float64:
                raise LnasrError("upload_gmmhmm_device takes float64 DeviceArrays of this context")
        N, M, D = mu.shape
        if A.shape != (N, N) or pi.shape != (N,) or w.shape != (N, M):
            raise ValueError("inconsistent GMM-HMM shapes")
        if sigma.shape != ((N, M, D) if diag else (N, M, D, D)):
            raise ValueError("sigma shape %s does not match (N,M,D)=%s" % (sigma.shape, (N, M, D)))
        if force_full and diag:
            raise ValueError("force_full needs a full covariance")
        self.model_owner = owner
        self._model_fp = None
        if force_full:
            check(lib().lnasr_model_upload_full(self.handle, N, M, D, A.ptr, pi.ptr, w.ptr, mu.ptr, sigma.ptr))
        else:
            check(lib().lnasr_model_upload(self.handle, N, M, D, A.ptr, pi.ptr, w.ptr, mu.ptr, sigma.ptr, 1 if diag else 0))
        self.model_shape = (N, M, D)
        self._model_fp = token

    def model_is_diag(self):
        v = C.c_int(0)
        check(lib().lnasr_model_is_diag(self.handle, C.byref(v)))
        return bool(v.value)

    def model_conditioning(self):
        """(kappa, direct_form): size of the terms the expanded-quadratic emission kernels cancel, and whether the
        model is therefore scored by the direct-form kernels (include/lnasr_hip.h)."""
        k, d = C.c_double(0), C.c_int(0)
        check(lib().lnasr_model_conditioning(self.handle, C.byref(k), C.byref(d)))
        return k.value, bool(d.value)

    def model_range(self):
        """(rho, integer_bank): the model's range as the integer matrix-core bank's fixed point sees it, and whether large
        batches of this model take that bank (include/lnasr_hip.h)."""
        r, d = C.c_double(0), C.c_int(0)
        check(lib().lnasr_model_range(self.handle, C.byref(r), C.byref(d)))
        return r.value, bool(d.value)

    # -- helpers ---------------------------------------------------------------------------
    def _out(self, like_device, shape, dtype):
        return DeviceArray(self, shape, dtype) if like_device else np.empty(shape, dtype=dtype)

    @staticmethod
    def _lens(lens, U):
        if lens is None:
            return None
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        if lens.shape != (U,):
            raise ValueError("lens must have shape (U,)")
        return lens

    # -- emission ----------------------------------------------------------------------------
    @staticmethod
    def _reuse(buf, shape, dtype):
        """`buf` if it is a DeviceArray of exactly this shape and dtype (an output buffer offered for reuse), else None."""
        if isinstance(buf, DeviceArray) and buf.ptr and buf.shape == tuple(shape) and buf.dtype == np.dtype(dtype):
            return buf
        return None

    def discrete_map_b(self, logB, obs, precision=np.float64, device_out=False, b_out=None):
        """device_out: leave b in HBM (a DeviceArray) although obs is a host array; b_out: a DeviceArray to write into
        when its shape and dtype fit (saves an allocation per call)."""
        logB = _in(logB, np.float64)
        obs = _in(obs, np.int32)
        if len(obs.shape) == 1:
            obs = obs.view((1, obs.shape[0])) if isinstance(obs, DeviceArray) else obs.reshape(1, -1)
        U, T = obs.shape
        N, K = logB.shape
        code = dtype_code(precision)
        b = self._reuse(b_out, (U, T, N), np_dtype(code))
        if b is None:
            b = self._out(device_out or isinstance(obs, DeviceArray), (U, T, N), np_dtype(code))
        check(lib().lnasr_discrete_map_b(self.handle, _ptr(logB), N, K, _ptr(obs), U, T, code, _ptr(b)))
        return b

    def gmm_logprob(self, obs, want_bm=False, precision=np.float64, lens=None, device_out=False, b_out=None, bm_out=None):
        """device_out: leave b / bm in HBM (DeviceArrays) although obs is a host array; b_out / bm_out: DeviceArrays to
        write into when their shapes and dtypes fit (saves two allocations per call)."""
        obs = _in(obs, np.float64)
        U, T, D = obs.shape
        N, M, D0 = self.model_shape
        if D != D0:
            raise ValueError("obs has D=%d, model has D=%d" % (D, D0))
        code = dtype_code(precision)
        dev = device_out or isinstance(obs, DeviceArray)
        b = self._reuse(b_out, (U, T, N), np_dtype(code))
        if b is None:
            b = self._out(dev, (U, T, N), np_dtype(code))
        bm = None
        if want_bm:
            bm = self._reuse(bm_out, (U, T, N, M), np_dtype(code))
            if bm is None:
                bm = self._out(dev, (U, T, N, M), np_dtype(code))
        check(lib().lnasr_gmm_logprob(self.handle, _ptr(obs), U, T, _ptr(self._lens(lens, U)), code, _ptr(b), _ptr(bm)))
        return (b, bm) if want_bm else b

    def gauss1d(self, x, mu, sigma2, want_log):
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty(x.shape, dtype=np.float64)
        check(lib().lnasr_gauss1d(self.handle, _ptr(x), x.size, float(mu), float(sigma2), int(want_log), _ptr(out)))
        return out

    def gauss_multi(self, x, mu, Sigma, want_log):
        x = np.ascontiguousarray(x, dtype=np.float64)
        L, D = x.shape
        mu = np.ascontiguousarray(np.asarray(mu, dtype=np.float64).reshape(D))
        Sigma = np.ascontiguousarray(np.asarray(Sigma, dtype=np.float64).reshape(D, D))
        out = np.empty(L, dtype=np.float64)
        check(lib().lnasr_gauss_multi(self.handle, _ptr(x), L, D, _ptr(mu), _ptr(Sigma), int(want_log), _ptr(out)))
        return out

    def gauss_mixture(self, c, x, mu, Sigma, want_log):
        x = np.ascontiguousarray(x, dtype=np.float64)
        L, D = x.shape
        c = np.ascontiguousarray(np.asarray(c, dtype=np.float64).ravel())
        M = c.shape[0]
        mu = np.ascontiguousarray(np.asarray(mu, dtype=np.float64).reshape(M, D))
        Sigma = np.ascontiguousarray(np.asarray(Sigma, dtype=np.float64).reshape(M, D, D))
        out = np.empty(L, dtype=np.float64)
        check(lib().lnasr_gauss_mixture(self.handle, _ptr(c), M, _ptr(x), L, D, _ptr(mu), _ptr(Sigma), int(want_log),
                                        _ptr(out)))
        return out

    # -- trellises -----------------------------------------------------------------------------
    def viterbi(self, b, want_v=True, want_bt=True, want_score=False, lens=None, host_out=False):
        """b [U,T,N] -> dict(path [U,T] u32, v, bt, score).  Outputs live where b lives; host_out: numpy arrays
        even for a device-resident b (the reference's methods keep b in HBM and return numpy)."""
        code = dtype_code(b.dtype)
        b = _in(b, np_dtype(code))
        U, T, N = b.shape
        dev = isinstance(b, DeviceArray) and not host_out
        v = self._out(dev, (U, T, N), np_dtype(code)) if want_v else None
        bt = self._out(dev, (U, T, N), np.uint32) if want_bt else None
        path = self._out(dev, (U, T), np.uint32)
        score = self._out(dev, (U,), np.float64) if want_score else None
        if bt is not None and not dev:
            bt[...] = 0
        check(lib().lnasr_viterbi(self.handle, _ptr(b), U, T, _ptr(self._lens(lens, U)), code, _ptr(v), _ptr(bt),
                                  _ptr(path), _ptr(score)))
        return {"path": path, "v": v, "bt": bt, "score": score}

    def decode(self, obs, want_score=False, lens=None, path_out=None, score_out=None):
        obs = _in(obs, np.float64)
        U, T, D = obs.shape
        dev = isinstance(obs, DeviceArray)
        path = path_out if path_out is not None else self._out(dev, (U, T), np.uint32)
        score = score_out if score_out is not None else (self._out(dev, (U,), np.float64) if want_score else None)
        check(lib().lnasr_decode(self.handle, _ptr(obs), U, T, _ptr(self._lens(lens, U)), _ptr(path), _ptr(score)))
        return (path, score) if (want_score or score_out is not None) else path

    def forward(self, b, want_alpha=True, want_loglik=True, lens=None, host_out=False, alpha_out=None):
        """alpha_out (and beta_out / xi_out / gamma_out below): a DeviceArray to write into when shape and dtype fit."""
        code = dtype_code(b.dtype)
        b = _in(b, np_dtype(code))
        U, T, N = b.shape
        dev = isinstance(b, DeviceArray) and not host_out
        alpha = None
        if want_alpha:
            alpha = self._reuse(alpha_out, (U, T, N), np_dtype(code)) if dev else None
            if alpha is None:
                alpha = self._out(dev, (U, T, N), np_dtype(code))
        ll = self._out(dev, (U,), np.float64) if want_loglik else None
        check(lib().lnasr_forward(self.handle, _ptr(b), U, T, _ptr(self._lens(lens, U)), code, _ptr(alpha), _ptr(ll)))
        return alpha, ll

    def backward(self, b, lens=None, host_out=False, beta_out=None):
        code = dtype_code(b.dtype)
        b = _in(b, np_dtype(code))
        U, T, N = b.shape
        dev = isinstance(b, DeviceArray) and not host_out
        beta = self._reuse(beta_out, (U, T, N), np_dtype(code)) if dev else None
        if beta is None:
            beta = self._out(dev, (U, T, N), np_dtype(code))
        check(lib().lnasr_backward(self.handle, _ptr(b), U, T, _ptr(self._lens(lens, U)), code, _ptr(beta)))
        return beta

    def calc_prob(self, obs, lens=None, precision=np.float64):
        """log P(O | model) per utterance.  precision=np.float32 scores the emission exactly as
        bw_accumulate(precision=np.float32) does, so both likelihoods of a training loop share one arithmetic."""
        obs = _in(obs, np.float64)
        U, T, D = obs.shape
        ll = self._out(isinstance(obs, DeviceArray), (U,), np.float64)
        check(lib().lnasr_calc_prob_p(self.handle, _ptr(obs), U, T, _ptr(self._lens(lens, U)), dtype_code(precision),
                                      _ptr(ll)))
        return ll

    def baumwelch(self, b, alpha, beta, want_xi=True, lens=None, host_out=False, xi_out=None, gamma_out=None):
        code = dtype_code(b.dtype)
        dt = np_dtype(code)
        b, alpha, beta = _in(b, dt), _in(alpha, dt), _in(beta, dt)
        U, T, N = b.shape
        dev = isinstance(b, DeviceArray) and not host_out
        xi = None
        if want_xi:
            xi = self._reuse(xi_out, (U, T, N, N), dt) if dev else None
            if xi is None:
                xi = self._out(dev, (U, T, N, N), dt)
        gamma = self._reuse(gamma_out, (U, T, N), dt) if dev else None
        if gamma is None:
            gamma = self._out(dev, (U, T, N), dt)
        check(lib().lnasr_baumwelch(self.handle, _ptr(b), _ptr(alpha), _ptr(beta), U, T, _ptr(self._lens(lens, U)),
                                    code, _ptr(xi), _ptr(gamma)))
        return xi, gamma

    def estimate_discrete(self, obs, K, xi, gamma, device_out=False, out=None):
        """device_out: the new A, B, pi stay in HBM as DeviceArrays (`out`: a dict of DeviceArrays to write into where shape
        and dtype fit)."""
        code = dtype_code(gamma.dtype)
        dt = np_dtype(code)
        obs = _in(obs, np.int32)
        xi, gamma = _in(xi, dt), _in(gamma, dt)
        T, N = gamma.shape
        res = {}
        for k, shp in (("A", (N, N)), ("B", (N, K)), ("pi", (N,))):
            buf = self._reuse((out or {}).get(k), shp, dt) if device_out else None
            res[k] = buf if buf is not None else self._out(device_out, shp, dt)
        A, B, pi = res["A"], res["B"], res["pi"]
        check(lib().lnasr_estimate_discrete(self.handle, _ptr(obs), T, N, K, code, _ptr(xi), _ptr(gamma), _ptr(A),
                                            _ptr(B), _ptr(pi)))
        return {"A": A, "B": B, "pi": pi}

    def estimate_gmm(self, obs, bm, alpha, beta, xi, gamma, min_std):
        """bm [T,N,M], alpha/beta/gamma [T,N], xi [T,N,N]; uses the uploaded (old) w and mu."""
        code = dtype_code(gamma.dtype)
        dt = np_dtype(code)
        obs = np.ascontiguousarray(obs, dtype=np.float64)
        bm, alpha, beta, xi, gamma = (_in(x, dt) for x in (bm, alpha, beta, xi, gamma))
        T, D = obs.shape
        N, M, _ = self.model_shape
        A = np.empty((N, N), dt)
        pi = np.empty(N, dt)
        w = np.empty((N, M), dt)
        mu = np.empty((N, M, D), np.float64)
        si = np.empty((N, M, D, D), np.float64)
        check(lib().lnasr_estimate_gmm(self.handle, _ptr(obs), T, code, _ptr(bm), _ptr(alpha), _ptr(beta), _ptr(xi),
                                       _ptr(gamma), float(min_std), _ptr(A), _ptr(pi), _ptr(w), _ptr(mu), _ptr(si)))
        return {"A": A, "pi": pi, "w": w, "mu": mu, "si": si}

    def bw_stats_len(self):
        n = C.c_size_t(0)
        check(lib().lnasr_bw_stats_len(self.handle, C.byref(n)))
        return n.value

    def bw_accumulate(self, obs, lens=None, stats_out=None, precision=np.float64):
        """Packed fp64 statistics of a batch.  precision=np.float32: densities and moments on the fp32 matrix cores
        (lnasr_bw_accumulate_p), statistics good to float32 accuracy."""
        obs = _in(obs, np.float64)
        U, T, D = obs.shape
        stats = stats_out if stats_out is not None else np.zeros(self.bw_stats_len(), dtype=np.float64)
        check(lib().lnasr_bw_accumulate_p(self.handle, _ptr(obs), U, T, _ptr(self._lens(lens, U)), dtype_code(precision),
                                          _ptr(stats)))
        return stats

    def bw_mstep(self, stats, min_std=0.01, device_out=False, out=None):
        """M-step on packed statistics (host array or DeviceArray).  device_out: the new parameters stay in HBM as
        DeviceArrays (`out`: a dict of DeviceArrays to write into where shape and dtype fit -- no allocation per call)."""
        N, M, D = self.model_shape
        diag = self.model_is_diag()
        sig = "var" if diag else "si"
        shapes = {"A": (N, N), "pi": (N,), "w": (N, M), "mu": (N, M, D), sig: (N, M, D) if diag else (N, M, D, D)}
        res = {}
        for k, shp in shapes.items():
            buf = self._reuse((out or {}).get("si" if k == "var" else k), shp, np.float64) if device_out else None
            res[k] = buf if buf is not None else self._out(device_out, shp, np.float64)
        stats = _in(stats, np.float64)
        check(lib().lnasr_bw_mstep(self.handle, _ptr(stats), float(min_std), _ptr(res["A"]), _ptr(res["pi"]), _ptr(res["w"]),
                                   _ptr(res["mu"]), _ptr(res[sig])))
        return res

    # -- numeric utils -------------------------------------------------------------------------
    def lse3(self, x, base2=False):
        """x [outer, n, inner] -> [outer, inner]."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        outer, n, inner = x.shape
        out = np.empty((outer, inner), dtype=np.float64)
        if out.size:
            check(lib().lnasr_lse(self.handle, _ptr(x), outer, n, inner, 1 if base2 else 0, _ptr(out)))
        return out

    def preemphasis(self, signal, alpha):
        signal = np.ascontiguousarray(signal, dtype=np.float64)
        out = np.empty(signal.shape, dtype=np.float64)
        check(lib().lnasr_preemphasis(self.handle, _ptr(signal), signal.size, float(alpha), _ptr(out)))
        return out

    def split_frames(self, signal, L, S):
        signal = np.ascontiguousarray(signal, dtype=np.float64)
        nf = num_frames(len(signal), L, S)
        out = np.empty((nf, L), dtype=np.float64)
        check(lib().lnasr_split_frames(self.handle, _ptr(signal), len(signal), int(L), int(S), _ptr(out)))
        return out

    def mfcc(self, signal, offsets, cfg, window, fbank, want_power=True, want_cepstrum=True, features_out=None):
        """signal: float64 or int16 samples (host or DeviceArray), utterances concatenated;
        offsets int64 [U+1].  Returns (power|None, cepstrum|None, features, frame_offsets).
        features_out: a DeviceArray [F, 3 (n_cep + 1)] to write into (saves a device allocation per call)."""
        if isinstance(signal, DeviceArray):
            kind = 1 if signal.dtype == np.int16 else 0
        else:
            signal = np.ascontiguousarray(signal)
            if signal.dtype == np.int16:
                kind = 1
            else:
                signal = np.ascontiguousarray(signal, dtype=np.float64)
                kind = 0
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        U = len(offsets) - 1
        # int(np.ceil(np.abs(n - (L - S)) / S)) per utterance (utils.py:119), as lnasr_num_frames computes it
        nfs = (np.abs(np.diff(offsets) - (cfg.frame_len - cfg.frame_step)) + cfg.frame_step - 1) // cfg.frame_step
        foff = np.concatenate([[0], np.cumsum(nfs)]).astype(np.int64)
        F = int(foff[-1])
        dev = isinstance(signal, DeviceArray)
        nbin = cfg.fft_n // 2 + 1
        power = self._out(dev, (F, nbin), np.float64) if want_power else None
        cep = self._out(dev, (F, cfg.n_mel), np.float64) if want_cepstrum else None
        if features_out is not None:
            if not isinstance(features_out, DeviceArray) or features_out.shape != (F, 3 * (cfg.n_cep + 1)) \
                    or features_out.dtype != np.float64:
                raise ValueError("features_out must be a float64 DeviceArray of shape (%d, %d)" % (F, 3 * (cfg.n_cep + 1)))
            feat = features_out
        else:
            feat = self._out(dev, (F, 3 * (cfg.n_cep + 1)), np.float64)
        if not isinstance(window, DeviceArray) and not isinstance(fbank, DeviceArray):
            # host tables: kept on the device between calls while their bytes stay the same (the reference
            # rebuilds the filter bank on every call, mfcc.py:78-106; the upload need not follow suit)
            window = np.ascontiguousarray(window, dtype=np.float64)
            fbank = np.ascontiguousarray(fbank, dtype=np.float64)
            fp = self._fingerprint("mfcc-tables", (window, fbank))
            cached = getattr(self, "_mfcc_tables", None)
            if fp is not None and cached is not None and cached[0] == fp:
                window, fbank = cached[1], cached[2]
            elif fp is not None:
                self._mfcc_tables = (fp, self.to_device(window), self.to_device(fbank))
                window, fbank = self._mfcc_tables[1], self._mfcc_tables[2]
        else:
            window, fbank = _in(window, np.float64), _in(fbank, np.float64)
        check(lib().lnasr_mfcc(self.handle, _ptr(signal), kind, _ptr(offsets), U, C.byref(cfg), _ptr(window),
                               _ptr(fbank), _ptr(power), _ptr(cep), _ptr(feat)))
        return power, cep, feat, foff

    def pad_rows(self, packed, row_offsets, pad_rows=None):
        """packed [F, W] (host or DeviceArray) + row offsets [U+1] -> ([U, pad_rows, W], lens [U] int32): the
        rectangular batch decode / calc_prob / bw_accumulate take.  pad_rows defaults to the longest utterance."""
        packed = _in(packed, np.float64)
        row_offsets = np.ascontiguousarray(row_offsets, dtype=np.int64)
        U, W = len(row_offsets) - 1, int(packed.shape[-1])
        lens = np.diff(row_offsets).astype(np.int32)
        if pad_rows is None:
            pad_rows = int(lens.max())
        out = self._out(isinstance(packed, DeviceArray), (U, int(pad_rows), W), np.float64)
        check(lib().lnasr_pad_rows(self.handle, _ptr(packed), _ptr(row_offsets), U, W, int(pad_rows), _ptr(out)))
        return out, lens

    # -- multi-GPU ---------------------------------------------------------------------------------
    def comm_init(self, unique_id, rank, nranks):
        buf = C.create_string_buffer(bytes(unique_id), UNIQUE_ID_BYTES)
        check(lib().lnasr_comm_init(self.handle, buf, int(rank), int(nranks)))
        self.rank, self.nranks = int(rank), int(nranks)

    def comm_info(self):
        """(rank, nranks) of the live communicator as RCCL reports them."""
        r, n = C.c_int(-1), C.c_int(0)
        check(lib().lnasr_comm_info(self.handle, C.byref(r), C.byref(n)))
        return r.value, n.value

    def comm_destroy(self):
        check(lib().lnasr_comm_destroy(self.handle))
        self.rank, self.nranks = 0, 1

    def allgather_u32(self, local, out=None):
        local = _in(local, np.uint32)
        n = int(np.prod(local.shape))
        nr = self.nranks
        if out is None:
            out = self._out(isinstance(local, DeviceArray), (nr,) + tuple(local.shape), np.uint32)
        check(lib().lnasr_allgather_u32(self.handle, _ptr(local), n, _ptr(out)))
        return out

    def allgather_paths(self, local, n_states, out=None):
        """allgather_u32 for decoded state paths: indices below 256 cross the links as bytes."""
        local = _in(local, np.uint32)
        n = int(np.prod(local.shape))
        if out is None:
            out = self._out(isinstance(local, DeviceArray), (self.nranks,) + tuple(local.shape), np.uint32)
        check(lib().lnasr_allgather_paths(self.handle, _ptr(local), n, int(n_states), _ptr(out)))
        return out

    def allreduce_f64(self, buf):
        if not isinstance(buf, DeviceArray):
            assert buf.dtype == np.float64 and buf.flags.c_contiguous
        n = int(np.prod(buf.shape))
        check(lib().lnasr_allreduce_f64(self.handle, _ptr(buf), n))
        return buf

    rank = 0
    nranks = 1


def comm_unique_id():
    buf = C.create_string_buffer(UNIQUE_ID_BYTES)
    check(lib().lnasr_comm_unique_id(buf))
    return buf.raw


def reload_switches():
    """Re-read the LNASR_* switches from the environment (the library reads them once per process)."""
    if _lib is not None or os.path.exists(_LIB_PATH):
        check(lib().lnasr_reload_switches())


def num_frames(n, L, S):
    out = C.c_int64(0)
    check(lib().lnasr_num_frames(int(n), int(L), int(S), C.byref(out)))
    return out.value


_default = None


def default_ctx():
    """The process-wide context (device = $LNASR_DEVICE or $LOCAL_RANK or 0)."""
    global _default
    if _default is None:
        _default = Context()
    return _default


def device_count():
    n = C.c_int(0)
    check(lib().lnasr_device_count(C.byref(n)))
    return n.value
