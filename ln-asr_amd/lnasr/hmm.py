"""Discrete-observation HMM with the reference's call surface (lnasr/hmm.py:16-313).

All probabilities (A, B, pi, b, alpha, beta, v, xi, gamma) are natural-log values; `A[i, j]`
is the transition i -> j.  Every trellis below runs in HIP kernels (csrc/viterbi.hip, backtrace.hip, fwdbwd.hip,
csrc/estimate.hip); this class only keeps the reference's state (`self.b`), shapes, dtypes and
print format.  Arrays returned are fresh numpy arrays in the reference's layouts:
`b [N,T]`, `alpha/beta/v [T,N]`, `bt [T,N] uint32`, `path [T] uint32`, `xi [T,N,N]`.

State: `_map_b` leaves `self.b` behind and the `remap=False` calls read it (hmm.py:82-83,95).  Here the
emission matrix stays in HBM after `_map_b` and the trellis methods read it there; `self.b` is a property that copies
it to a numpy array `[N,T]` on first read.  From that moment -- or when the caller assigns an array of its own
(`h.b = ...`) -- the host array is the authoritative one (the caller may edit it in place) and the next trellis call
sends it to the device again.

Extras (keyword-only / new methods, not in the reference): `ctx=` selects the GPU context,
`decode_batch` / `calc_prob_batch` take `[U, T]` symbol batches.
"""
import contextlib
import itertools
from typing import Tuple

import numpy as np

from . import _ffi
from .utils import lse


def _same(x):
    return x


_SERIAL = itertools.count(1)          # one number per device-resident parameter set, process-wide (never reused)


class HMM:
    """n hidden states, m observation symbols, log-domain parameters (hmm.py:17-45)."""

    def __init__(self, n: int = 1, m: int = 1, A=None, B=None, pi=None, precision=np.double, *, ctx=None):
        self.n = n
        self.m = m
        self.A = A
        self.B = B
        self.pi = pi
        self.precision = precision
        self._ctx = ctx

    # ------------------------------------------------------------------ plumbing
    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = _ffi.default_ctx()
        return self._ctx

    def _upload(self):
        """Transition model -> device.  Checked per call (the reference lets callers mutate A / pi in place between
        calls); the device keeps what it has when the fingerprint is unchanged.  A and pi an on-device M-step left in HBM
        (`train`) are installed device to device, unchecked: nobody outside this object has seen them."""
        if self._params_current():
            return
        ctx = self.ctx
        res = self.__dict__.get("_resident", {})
        if "A" in res or "pi" in res:
            if all(k in res and res[k][0].ctx is ctx and res[k][0].ptr for k in ("A", "pi")):
                token = ("dev-hmm", self._dev_serial)
                if ctx._model_fp != token:
                    ctx.upload_hmm_device(res["A"][0], res["pi"][0], token=token)
                return
        ctx.upload_hmm(self.A, self.pi)                    # (reading a resident one makes it a host array)

    @contextlib.contextmanager
    def _one_check(self, enabled):
        """Inside one outer call (a `train` iteration) whose steps are all this package's own, nothing touches the
        parameters between the first step and `_updatemodel`: they are fingerprinted once, by the first step that
        needs them, instead of once per inner step."""
        self._check_state = "armed" if enabled else None
        try:
            yield
        finally:
            self._check_state = None

    def _params_current(self):
        st = self.__dict__.get("_check_state")
        if st == "armed":
            self._check_state = "done"
        return st == "done"

    def _remap_or_upload(self, obs, remap):
        """Start of every trellis call: optionally recompute self.b, and make sure the device holds the current
        parameters.  When `_map_b` has just uploaded (and fingerprinted) the whole model inside this very call
        (GMMHMM does), no user code ran in between and the second check is skipped."""
        if remap:
            self._model_fresh = False
            self._map_b(obs)
            if getattr(self, "_model_fresh", False):
                self._model_fresh = False
                return
        self._upload()

    # `b` (and GMMHMM's `bm`): device-resident after _map_b, a numpy array once somebody looks at it or assigns one
    def _keep(self, name, dev, to_reference_layout):
        self.__dict__.setdefault("_resident", {})[name] = (dev, to_reference_layout)
        self.__dict__.setdefault("_host", {}).pop(name, None)
        self.__dict__.setdefault("_spare", {}).pop(name, None)

    def _device_buffer(self, name):
        """The device array the previous _map_b wrote `name` into, offered to the next one as its output buffer (same
        utterance length: no allocation per call).  Nobody outside holds it: readers get host copies."""
        res = self.__dict__.get("_resident", {})
        return res[name][0] if name in res else self.__dict__.get("_spare", {}).get(name)

    def _host_get(self, name):
        res = self.__dict__.get("_resident", {})
        host = self.__dict__.setdefault("_host", {})
        if name in res:
            dev, layout = res.pop(name)          # handed out = the caller may write into it: the host copy rules
            host[name] = layout(dev.to_host())
            self.__dict__.setdefault("_spare", {})[name] = dev
        try:
            return host[name]
        except KeyError:
            raise AttributeError("'%s' object has no attribute '%s' (call _map_b first)" % (type(self).__name__, name))

    def _host_set(self, name, value):
        self.__dict__.setdefault("_resident", {}).pop(name, None)
        self.__dict__.setdefault("_host", {})[name] = value

    b = property(lambda self: self._host_get("b"), lambda self, value: self._host_set("b", value),
                 doc="emission log-probabilities [N,T] of the last _map_b (hmm.py:58-61)")

    # A, B, pi are numpy arrays as in the reference -- except right after the on-device M-step of `train`, which leaves the
    # NEW parameters in HBM for the next iteration.  Same mechanism as `b`: reading one copies it to a numpy array, and from
    # then on -- or when the caller assigns an array -- the host array is the parameter (it may be edited in place) and goes
    # through the fingerprinted upload again.
    _PARAMS = ("A", "B", "pi")
    A = property(lambda self: self._host_get("A"), lambda self, value: self._host_set("A", value))
    B = property(lambda self: self._host_get("B"), lambda self, value: self._host_set("B", value))
    pi = property(lambda self: self._host_get("pi"), lambda self, value: self._host_set("pi", value))

    def _keep_param(self, name, dev):
        """A parameter the device produced: resident until somebody reads the attribute; the array it replaces is kept as
        the output buffer of the next M-step (two buffers per parameter, no allocation per iteration)."""
        res = self.__dict__.setdefault("_resident", {})
        spare = self.__dict__.setdefault("_spare", {})
        old = res.pop(name, None)
        if old is not None and old[0] is not dev:
            spare[name] = old[0]
        elif spare.get(name) is dev:
            spare.pop(name)
        res[name] = (dev, _same)
        self.__dict__.setdefault("_host", {}).pop(name, None)

    def _reuse_buffer(self, name, shape, dtype=np.float64):
        buf = self.__dict__.get(name)
        if buf is None or buf.ctx is not self.ctx or not buf.ptr or buf.shape != tuple(shape) or buf.dtype != np.dtype(dtype):
            buf = self.ctx.empty(shape, dtype)
            self.__dict__[name] = buf
        return buf

    def _b_dev(self):
        """The emissions as the kernels take them, frame-major [1,T,N]: the array _map_b left in HBM, or the host
        array (the caller's, or one already read through `self.b`) in the storage dtype."""
        res = self.__dict__.get("_resident", {})
        if "b" in res:
            return res["b"][0]
        b = np.asarray(self.b)
        dt = np.float32 if b.dtype == np.float32 else np.float64
        return np.ascontiguousarray(b.T, dtype=dt)[None]

    def _ours(self, *names):
        """True when none of the named methods is replaced by a subclass: the fused routes (decode, calc_prob, the
        on-device training step) compute exactly what the reference's sequence of calls does, and are taken only then."""
        for name in names:
            owner = next(c for c in type(self).__mro__ if name in c.__dict__)
            if owner.__module__.split(".")[0] != __name__.split(".")[0]:
                return False
        return True

    # ------------------------------------------------------------------ emission
    def _map_b(self, obs: np.ndarray):
        """b[:, t] = B[:, obs[t]]  (hmm.py:47-61)."""
        obs = np.asarray(obs)
        b = self.ctx.discrete_map_b(self._B_resident(), obs.astype(np.int32), self.precision,
                                    device_out=True, b_out=self._device_buffer("b"))
        self._keep("b", b, lambda h: h[0].T)

    def _B_resident(self):
        """The emission table in HBM, sent again only when its bytes changed (scripts/seg/seg.py's B is [4, 65536]: 2 MB per
        `decode` otherwise; the check is the same fingerprint the transition model gets)."""
        res = self.__dict__.get("_resident", {})
        if "B" in res and res["B"][0].ctx is self.ctx and res["B"][0].ptr:
            return res["B"][0]                              # the table the M-step of `train` left there
        B = np.asarray(self.B)
        fp = self.ctx._fingerprint("B", (B,))
        held = self.__dict__.get("_B_dev")
        if held is None or held[0] != fp or held[1].ctx is not self.ctx or not held[1].ptr:
            held = (fp, self.ctx.to_device(B, dtype=np.float64))
            self._B_dev = held
        return held[1]

    # ------------------------------------------------------------------ trellises
    def _forward(self, obs: np.ndarray, remap: bool = False) -> np.ndarray:
        """alpha [T,N]: alpha[t,j] = lse_i(alpha[t-1,i] + A[i,j]) + b[j,t]  (hmm.py:63-100)."""
        def run():
            self._remap_or_upload(obs, remap)
            return self.ctx.forward(self._b_dev(), want_alpha=True, want_loglik=False, host_out=True)[0][0]
        return self._outer(run)

    def _backward(self, obs: np.ndarray, remap: bool = False) -> np.ndarray:
        """beta [T,N]: beta[t,i] = lse_j(beta[t+1,j] + A[i,j] + b[j,t+1])  (hmm.py:102-131)."""
        def run():
            self._remap_or_upload(obs, remap)
            return self.ctx.backward(self._b_dev(), host_out=True)[0]
        return self._outer(run)

    def _viterbi(self, obs: np.ndarray, remap: bool = False) -> Tuple[np.ndarray]:
        """(v [T,N], bt [T,N] uint32, path [T] uint32); ties -> first index  (hmm.py:133-172)."""
        def run():
            self._remap_or_upload(obs, remap)
            r = self.ctx.viterbi(self._b_dev(), want_v=True, want_bt=True, host_out=True)
            return (r["v"][0], r["bt"][0], r["path"][0])
        return self._outer(run)

    def _baumwelch(self, obs: np.ndarray, alpha: np.ndarray, beta: np.ndarray, remap: bool = False) -> Tuple[np.ndarray]:
        """(xi [T,N,N], gamma [T,N]); the last frame is -inf  (hmm.py:174-200)."""
        self._remap_or_upload(obs, remap)
        b = self._b_dev()
        dt = b.dtype
        xi, gamma = self.ctx.baumwelch(b, np.asarray(alpha, dtype=dt)[None], np.asarray(beta, dtype=dt)[None],
                                       host_out=True)
        return xi[0], gamma[0]

    def _estimate(self, obs: np.ndarray, alpha, beta, xi, gamma) -> dict:
        """M-step {'A','B','pi'}; a symbol never observed gets a -inf column  (hmm.py:202-225)."""
        dt = np.float32 if np.asarray(gamma).dtype == np.float32 else np.float64
        return self.ctx.estimate_discrete(np.asarray(obs), self.m, np.asarray(xi, dtype=dt), np.asarray(gamma, dtype=dt))

    def _updatemodel(self, model: dict):
        for k in ("A", "B", "pi"):
            if isinstance(model[k], _ffi.DeviceArray):
                self._keep_param(k, model[k])
            else:
                setattr(self, k, model[k])
        self._dev_serial = next(_SERIAL)

    def reset(self, init_type):
        """'uniform' or 'random' initial parameters; anything else is a silent no-op (hmm.py:233-250).
        Uses the global numpy RNG in the reference's draw order (A, then B, then pi), so seeding reproduces it."""
        shapes = {"A": (self.n, self.n), "B": (self.n, self.m), "pi": (self.n,)}
        if init_type == "uniform":
            for name, shape in shapes.items():
                setattr(self, name, _log_uniform(shape, self.precision))
        elif init_type == "random":
            # 1 - U[0, 1) lies in (0, 1]: never log(0)
            for name, shape in shapes.items():
                setattr(self, name, _log_normalised(1.0 - np.random.random_sample(shape)))

    # ------------------------------------------------------------------ persistence
    def save(self, filename):
        """Datasets 'A','B','pi' (hmm.py:252-258): HDF5 when h5py is importable, else .npz."""
        _save(filename, {"A": self.A, "B": self.B, "pi": self.pi})

    def load(self, filename):
        """hmm.py:260-266; unlike the reference the parameters come back as real ndarrays."""
        d = _load(filename, ("A", "B", "pi"))
        self.A, self.B, self.pi = d["A"], d["B"], d["pi"]
        self.n, self.m = self.B.shape

    # ------------------------------------------------------------------ public API
    def _confirmed(self):
        """End of a speculative attempt (GMMHMM._upload_gmm): True when no background check is pending or it confirms that
        the device held exactly the current parameters."""
        job = self.__dict__.pop("_spec_job", None)
        return job is None or job.result() == self.ctx._model_fp

    def _outer(self, fn):
        """Run `fn` (emission + one trellis, the answer only) first on the model the device already holds, with the
        parameter check beside it, and again the checked way if the check says the parameters have changed."""
        self._speculating = True
        first_error = None
        try:
            out = fn()
        except Exception as e:                          # noqa: BLE001
            self._speculating = False
            if self._confirmed():
                raise                                   # not a stale-model artefact: the caller's error
            out, first_error = None, e
        finally:
            self._speculating = False
        if out is not None and self._confirmed():
            return out
        try:
            return fn()
        except Exception as e:                          # noqa: BLE001
            if first_error is not None:                 # both attempts failed: say what the first one died of as well
                raise e from first_error
            raise

    def calc_prob(self, obs: np.ndarray) -> float:
        """log P(O | model) = lse(alpha[T-1])  (hmm.py:268-274).  alpha itself is not brought back unless a subclass
        replaced `_forward`."""
        if not self._ours("_forward"):
            return lse(self._forward(obs, True)[-1])

        def run():
            self._remap_or_upload(obs, True)
            _, ll = self.ctx.forward(self._b_dev(), want_alpha=False, want_loglik=True, host_out=True)
            return np.asarray(ll, dtype=self.precision)[0]
        return self._outer(run)

    def decode(self, obs: np.ndarray) -> np.ndarray:
        """Most likely state sequence  (hmm.py:276-283): emission, trellis and back-trace on the device, only the path
        [T] comes back (v and bt, which the reference computes and drops here, are not copied; `self.b` -- and
        GMMHMM's `self.bm` -- stay readable as after the reference's call)."""
        if not self._ours("_viterbi"):
            return self._viterbi(obs, True)[2]

        def run():
            self._remap_or_upload(obs, True)
            return self.ctx.viterbi(self._b_dev(), want_v=False, want_bt=False, host_out=True)["path"][0]
        return self._outer(run)

    _REF_STEPS = ("_map_b", "_forward", "_backward", "_baumwelch", "_estimate", "_updatemodel", "calc_prob")

    def train(self, obs, iters=1, eps=0.0001, verbose: bool = True):
        """Baum-Welch on one sequence, stopping when |delta loglik| < eps  (hmm.py:285-313).  The E and M steps go
        through the overridable methods in the reference's order, so subclasses that replace one of them keep working."""
        ours = self._ours(*self._REF_STEPS)
        if ours and type(self)._estimate is HMM._estimate:
            if self.precision in (np.double, np.float64):
                return self._train_fused(obs, iters, eps, verbose)
            return self._train_resident(obs, iters, eps, verbose)
        for k in range(iters):
            with self._one_check(ours):
                self._map_b(obs)
                alpha, beta = self._forward(obs), self._backward(obs)
                posteriors = self._baumwelch(obs, alpha, beta)
                new_model = self._estimate(obs, alpha, beta, *posteriors)
            before = lse(alpha[-1])
            self._updatemodel(new_model)
            after = self.calc_prob(obs)
            if _report_iteration(k, before, after, verbose) < eps:
                break

    def _train_fused(self, obs, iters, eps, verbose):
        """float64, no step overridden: the whole loop in HBM.  The observations go up once; alpha, beta, xi [T,N,N], gamma and
        the M-step's new A, B, pi stay there (the parameters are installed as the next model device to device and become numpy
        arrays when somebody reads the attributes); the buffers are allocated once.  Per iteration ONE row crosses PCIe:
        alpha[T-1], for the reference's `lse(alpha[-1])`.  And L(lambda_new | O) of iteration k is L(lambda | O) of
        iteration k + 1 -- the reference runs that forward pass twice over the same model (`calc_prob` at hmm.py:304, then
        `_forward` at :295), here it runs once.  `self.b` ends as the reference's closing `calc_prob` leaves it: the final
        model's emissions.  test/hmm-test.py's 48 iterations: 0.65 -> 0.3 ms each."""
        if iters <= 0:
            return                                           # the reference's `for k in range(iters)` does nothing
        ctx = self.ctx
        obs_i = np.ascontiguousarray(obs, dtype=np.int32)
        T, N = obs_i.shape[0], self.n
        d_obs = ctx.to_device(obs_i[None])
        try:
            self._train_fused_loop(d_obs, T, N, iters, eps, verbose)
        finally:
            d_obs.free()                                     # also when a kernel call raises or the user interrupts

    def _train_fused_loop(self, d_obs, T, N, iters, eps, verbose):
        ctx = self.ctx
        buf = {k: self._reuse_buffer("_train_" + k, shp)
               for k, shp in (("alpha", (1, T, N)), ("beta", (1, T, N)), ("xi", (1, T, N, N)), ("gamma", (1, T, N)))}

        def forward_pass():                                  # `_map_b` + `_forward` of the current model -> lse(alpha[-1])
            self._upload()
            b = ctx.discrete_map_b(self._B_resident(), d_obs, self.precision, device_out=True, b_out=self._device_buffer("b"))
            self._keep("b", b, lambda h: h[0].T)
            ctx.forward(b, want_alpha=True, want_loglik=False, alpha_out=buf["alpha"])
            return b, lse(buf["alpha"].rows_to_host(T - 1, 1, N)[0])

        b, before = forward_pass()
        for k in range(iters):
            ctx.backward(b, beta_out=buf["beta"])
            ctx.baumwelch(b, buf["alpha"], buf["beta"], xi_out=buf["xi"], gamma_out=buf["gamma"])
            spare = self.__dict__.get("_spare", {})
            self._updatemodel(ctx.estimate_discrete(d_obs.view((T,)), self.m, buf["xi"].view((T, N, N)),
                                                    buf["gamma"].view((T, N)), device_out=True,
                                                    out={p: spare.get(p) for p in self._PARAMS}))
            b, after = forward_pass()
            if _report_iteration(k, before, after, verbose) < eps:
                break
            before = after

    def _train_resident(self, obs, iters, eps, verbose):
        """The same iteration with every intermediate left in HBM (taken when no step is overridden): `_map_b`, forward, backward,
        posteriors and the discrete M-step are the calls `train` makes through the overridable methods, but alpha, beta,
        xi [T,N,N] and gamma go from kernel to kernel as device arrays -- only alpha[T-1] (for the reference's `lse(alpha[-1])`)
        and the new A, B, pi come back.  Same kernels, same arithmetic, same printed line; test/hmm-test.py's 48 iterations:
        1.6 -> 0.6 ms each."""
        ctx = self.ctx
        obs_i = np.ascontiguousarray(obs, dtype=np.int32)
        for k in range(iters):
            with self._one_check(True):
                self._map_b(obs)
                self._upload()
                b = self._b_dev()                                       # [1,T,N] in HBM
                T, N = b.shape[1], b.shape[2]
                alpha, _ = ctx.forward(b, want_alpha=True, want_loglik=False)
                beta = ctx.backward(b)
                xi, gamma = ctx.baumwelch(b, alpha, beta)
                new_model = ctx.estimate_discrete(obs_i, self.m, xi.view((T, N, N)), gamma.view((T, N)))
                before = lse(alpha.rows_to_host(T - 1, 1, N)[0])        # the reference's lse(alpha[-1]) on the same values
            self._updatemodel(new_model)
            after = self.calc_prob(obs)
            if _report_iteration(k, before, after, verbose) < eps:
                break

    # ------------------------------------------------------------------ batch extras
    def decode_batch(self, obs_batch):
        """[U,T] int symbols -> [U,T] uint32 paths."""
        obs_batch = np.asarray(obs_batch, dtype=np.int32)
        self._upload()
        b = self.ctx.discrete_map_b(np.asarray(self.B, dtype=np.float64), obs_batch, self.precision)
        return self.ctx.viterbi(b, want_v=False, want_bt=False)["path"]

    def calc_prob_batch(self, obs_batch):
        obs_batch = np.asarray(obs_batch, dtype=np.int32)
        self._upload()
        b = self.ctx.discrete_map_b(np.asarray(self.B, dtype=np.float64), obs_batch, self.precision)
        return self.ctx.forward(b, want_alpha=False, want_loglik=True)[1]


_ITER_LINE = "Iter: {:3}, L(lambda|O) = {:.6e}, L(lambda_new|O) = {:.6e}, eps = {:.6f}"


def _report_iteration(k, before, after, verbose):
    """|after - before|, printing the reference's progress line (hmm.py:306-311; the text is observable behaviour)."""
    delta = abs(before - after)
    if verbose:
        print(_ITER_LINE.format(k, before, after, delta))
    return delta


def _log_uniform(shape, dtype):
    """log of a uniform distribution over the last axis, in `dtype` like the reference's np.ones(..., dtype) / n."""
    return np.log(np.ones(shape, dtype=dtype) * (1.0 / shape[-1]))


def _log_normalised(draw):
    """log of `draw` scaled so that its last axis sums to one (stays float64 whatever `precision`, as in the reference)."""
    return np.log(draw / draw.sum(axis=-1, keepdims=True))


def _save(filename, arrays):
    try:
        import h5py
    except ImportError:
        h5py = None
    if h5py is not None and not str(filename).endswith(".npz"):
        with h5py.File(filename, "w") as f:
            for k, v in arrays.items():
                f.create_dataset(k, data=np.asarray(v))
        return
    with open(filename, "wb") as fp:
        np.savez(fp, **{k: np.asarray(v) for k, v in arrays.items()})


def _load(filename, keys):
    with open(filename, "rb") as fp:
        magic = fp.read(8)
    if magic.startswith(b"\x89HDF"):
        import h5py
        with h5py.File(filename, "r") as f:
            return {k: np.array(f[k]) for k in keys}
    with np.load(filename) as z:
        return {k: np.array(z[k]) for k in keys}
