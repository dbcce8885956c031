"""Runs inside the sanitizer process of tests/test_host_asan.py (LNASR_LIB = the --host-asan build, ASan runtime
preloaded): every no-device path of the C-ABI's host side -- library load and symbol table, error plumbing
(lnasr_last_error, LN_REQUIRE / fail), argument checks with null contexts and null pointers, lnasr_num_frames,
the switch table reload, context creation without a device.  Any heap / stack error or undefined behaviour aborts
the process (ASAN_OPTIONS=abort_on_error=1, -fno-sanitize-recover=undefined)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ln-asr_amd"))
from lnasr import _ffi  # noqa: E402

lib = _ffi.lib()
assert os.path.basename(_ffi._LIB_PATH) == "liblnasr_hip_asan.so", _ffi._LIB_PATH
assert lib.lnasr_abi_version() == 1
checked = 0

# frame-count helper: pure host arithmetic (utils.py:119), including the degenerate sizes
for n, L, S, want in ((30, 3, 2, 15), (160000, 400, 160, 999), (160240, 400, 160, 1000), (100, 400, 160, 1), (400, 400, 160, 1)):
    assert _ffi.num_frames(n, L, S) == want, (n, L, S)
    checked += 1
out = C.c_int64(0)
assert lib.lnasr_num_frames(10, 0, 0, C.byref(out)) < 0 and lib.lnasr_last_error()      # bad arguments: error, no crash
assert lib.lnasr_num_frames(10, 4, 2, None) < 0

# switches: loaded once, reloadable, with every kind of value in the environment
os.environ.update(LNASR_NO_MFMA="1", LNASR_VIT="a", LNASR_KAPPA_MAX="17.5", LNASR_FRAMEPAR_MAXU="3", LNASR_H2D_THREADS="99")
assert lib.lnasr_reload_switches() == 0
for k in ("LNASR_NO_MFMA", "LNASR_VIT", "LNASR_KAPPA_MAX", "LNASR_FRAMEPAR_MAXU", "LNASR_H2D_THREADS"):
    os.environ.pop(k)
os.environ["LNASR_H2D_THREADS"] = "-4"
assert lib.lnasr_reload_switches() == 0
os.environ.pop("LNASR_H2D_THREADS")
assert lib.lnasr_reload_switches() == 0

# parameter fingerprint (host only): every tail length, the threaded path (>= 2 MB), sensitivity to one bit, null checks
import numpy as np  # noqa: E402
rng = np.random.default_rng(0)
o1, o2 = (C.c_uint64 * 2)(), (C.c_uint64 * 2)()
for nbytes in (0, 1, 7, 8, 9, 31, 32, 33, 63, 64, 1000, (1 << 21) - 1, (1 << 21) + 5, 25 * (1 << 20) + 3):
    x = rng.integers(0, 256, nbytes, dtype=np.uint8)
    assert lib.lnasr_hash128(x.ctypes.data if nbytes else None, nbytes, 7, o1) == 0
    assert lib.lnasr_hash128(x.ctypes.data if nbytes else None, nbytes, 7, o2) == 0 and tuple(o1) == tuple(o2)
    assert lib.lnasr_hash128(x.ctypes.data if nbytes else None, nbytes, 8, o2) == 0 and tuple(o1) != tuple(o2)   # seed
    for at in ({0, nbytes // 2, nbytes - 1} if nbytes else ()):
        y = x.copy()
        y[at] ^= 1
        assert lib.lnasr_hash128(y.ctypes.data, nbytes, 7, o2) == 0 and tuple(o1) != tuple(o2), (nbytes, at)
    checked += 1
assert lib.lnasr_hash128(None, 16, 0, o1) < 0 and lib.lnasr_hash128(None, 0, 0, None) < 0
# the background job: same digests as the synchronous calls (buffer k with seed + k), empty buffers, discarded results
bufs = [rng.integers(0, 256, n, dtype=np.uint8) for n in (3 << 20, 0, 17, (5 << 20) + 1)]
job = _ffi._FingerprintJob("head", bufs)
want = ["head"]
for k, x in enumerate(bufs):
    if x.size:
        assert lib.lnasr_hash128(x.ctypes.data, x.nbytes, k, o1) == 0
        want.append((o1[0], o1[1]))
assert job.result() == tuple(want)
for _ in range(20):
    del job
    job = _ffi._FingerprintJob("head", bufs)                      # never collected: ended by the destructor
del job
h2 = C.c_void_p()
assert lib.lnasr_hash128_begin(None, None, 2, 0, C.byref(h2)) < 0 and not h2.value
assert lib.lnasr_hash128_begin(None, None, 0, 0, None) < 0 and lib.lnasr_hash128_end(None, None) < 0
checked += 3

# the launch helpers' pointer guard (LN_DEVPTR, always on in this build): a heap pointer -- what a numpy array hands over --
# is refused with a message that names it; null (an optional output) passes
heap = np.zeros(64)
rc = lib.lnasr_debug_require_device_ptr(heap.ctypes.data)
assert rc == -1 and b"not device memory" in lib.lnasr_last_error(), (rc, lib.lnasr_last_error())
stack = (C.c_double * 8)()
assert lib.lnasr_debug_require_device_ptr(C.addressof(stack)) == -1
assert lib.lnasr_debug_require_device_ptr(None) == 0
checked += 3

# device count / context creation without a device: an error code and a message, never a crash
n = C.c_int(-1)
assert lib.lnasr_device_count(C.byref(n)) == 0
assert lib.lnasr_device_count(None) < 0
have_gpu = n.value > 0
h = C.c_void_p()
if not have_gpu:
    rc = lib.lnasr_ctx_create(0, C.byref(h))
    assert rc < 0 and b"no HIP device" in lib.lnasr_last_error(), lib.lnasr_last_error()
    assert not h.value
assert lib.lnasr_ctx_create(0, None) < 0
lib.lnasr_ctx_destroy(None)                                      # documented no-op

# every entry point that takes a context refuses a null one (LN_REQUIRE -> LNASR_E_INVALID + message)
null = None
buf = (C.c_double * 64)()
ibuf = (C.c_int32 * 64)()
u32 = (C.c_uint32 * 64)()
name = C.create_string_buffer(32)
cfg = _ffi.MfccCfgStruct(16000, 400, 160, 512, 40, 12, 0.97)
szt = C.c_size_t(0)
i64 = C.c_int64(0)
dbl = C.c_double(0)
ci = C.c_int(0)
off = (C.c_int64 * 2)(0, 16)
calls = {
    "lnasr_sync": (null,), "lnasr_device_sync": (null,), "lnasr_device_info": (null, name, 32, C.byref(ci), C.byref(szt)),
    "lnasr_dev_alloc": (null, 16, C.byref(h)), "lnasr_dev_free": (null, null), "lnasr_host_alloc": (null, 16, C.byref(h)),
"lnasr_memcpy_h2d": (null, buf, buf, 8), "lnasr_memcpy_d2h": (null, buf, buf, 8),
    "lnasr_memset": (null, buf, 0, 8), "lnasr_prof_enable": (null, 1), "lnasr_prof_reset": (null,),
    "lnasr_prof_count": (null, C.byref(ci)), "lnasr_prof_get": (null, 0, name, 32, C.byref(i64), C.byref(dbl)),
    "lnasr_model_upload": (null, 2, 2, 2, buf, buf, buf, buf, buf, 1),
    "lnasr_model_upload_full": (null, 2, 2, 2, buf, buf, buf, buf, buf), "lnasr_hmm_upload": (null, 2, buf, buf),
    "lnasr_model_is_diag": (null, C.byref(ci)), "lnasr_model_conditioning": (null, C.byref(dbl), C.byref(ci)),
    "lnasr_model_range": (null, C.byref(dbl), C.byref(ci)),
    "lnasr_discrete_map_b": (null, buf, 2, 2, ibuf, 1, 2, 0, buf), "lnasr_gmm_logprob": (null, buf, 1, 2, null, 0, buf, null),
    "lnasr_gauss1d": (null, buf, 4, 0.0, 1.0, 1, buf), "lnasr_gauss_multi": (null, buf, 2, 2, buf, buf, 1, buf),
    "lnasr_gauss_mixture": (null, buf, 2, buf, 2, 2, buf, buf, 1, buf),
    "lnasr_viterbi": (null, buf, 1, 2, null, 0, null, null, u32, null), "lnasr_decode": (null, buf, 1, 2, null, u32, null),
    "lnasr_forward": (null, buf, 1, 2, null, 0, buf, buf), "lnasr_backward": (null, buf, 1, 2, null, 0, buf),
    "lnasr_calc_prob": (null, buf, 1, 2, null, buf), "lnasr_calc_prob_p": (null, buf, 1, 2, null, 1, buf),
    "lnasr_baumwelch": (null, buf, buf, buf, 1, 2, null, 0, buf, buf),
    "lnasr_estimate_discrete": (null, ibuf, 2, 2, 2, 0, buf, buf, buf, buf, buf),
    "lnasr_estimate_gmm": (null, buf, 2, 0, buf, buf, buf, buf, buf, 0.01, buf, buf, buf, buf, buf),
    "lnasr_bw_stats_len": (null, C.byref(szt)), "lnasr_bw_accumulate": (null, buf, 1, 2, null, buf),
    "lnasr_bw_accumulate_p": (null, buf, 1, 2, null, 1, buf), "lnasr_bw_mstep": (null, buf, 0.01, buf, buf, buf, buf, buf),
    "lnasr_lse": (null, buf, 1, 2, 1, 0, buf), "lnasr_preemphasis": (null, buf, 8, 0.97, buf),
    "lnasr_split_frames": (null, buf, 8, 4, 2, buf),
    "lnasr_mfcc": (null, buf, 0, off, 1, C.byref(cfg), buf, buf, null, null, buf),
    "lnasr_pad_rows": (null, buf, off, 1, 2, 8, buf), "lnasr_comm_init": (null, buf, 0, 1),
    "lnasr_comm_info": (null, C.byref(ci), C.byref(ci)), "lnasr_allgather_u32": (null, u32, 4, u32),
    "lnasr_allgather_paths": (null, u32, 4, 8, u32), "lnasr_allreduce_f64": (null, buf, 4),
    "lnasr_debug_fallback_columns": (null, u32), "lnasr_debug_clock": (null, C.byref(dbl)),
}
no_ctx = {"lnasr_abi_version", "lnasr_last_error", "lnasr_device_count", "lnasr_ctx_create", "lnasr_ctx_destroy",
          "lnasr_num_frames", "lnasr_reload_switches", "lnasr_comm_unique_id", "lnasr_comm_destroy", "lnasr_hash128",
          "lnasr_debug_require_device_ptr", "lnasr_host_free", "lnasr_hash128_begin", "lnasr_hash128_end"}
missing = sorted(set(_ffi.SIGNATURES) - set(calls) - no_ctx)
assert not missing, "entry points without a null-context call here: %s" % missing
for fn, args in sorted(calls.items()):
    rc = getattr(lib, fn)(*args)
    assert rc < 0, "%s accepted a null context (rc=%d)" % (fn, rc)
    assert lib.lnasr_last_error(), fn
    checked += 1
assert lib.lnasr_comm_destroy(None) == 0                         # documented no-op
assert lib.lnasr_host_free(None, None) == 0                      # null block: no-op whatever the context
assert lib.lnasr_comm_unique_id(None) < 0

# the Python layer turns the codes into exceptions that carry the message
try:
    _ffi.check(lib.lnasr_sync(None))
    raise SystemExit("no exception")
except _ffi.LnasrError as e:
    assert "null ctx" in str(e)
if not have_gpu:
    try:
        _ffi.Context(0)
        raise SystemExit("Context() succeeded without a device")
    except _ffi.LnasrError as e:
        assert "no HIP device" in str(e)
print("asan host paths ok: %d checks" % checked)
