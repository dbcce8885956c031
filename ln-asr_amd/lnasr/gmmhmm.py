"""GMM-HMM with the reference's call surface (lnasr/gmmhmm.py:17-208).

n states, m Gaussians per state, d-dimensional observations; `A, pi, w` are natural-log
probabilities, `mu [N,M,D]`, `si [N,M,D,D]` linear.  Emission scoring, the trellises and the
Baum-Welch M-step all run in HIP kernels; `_map_b` leaves `self.b [N,T]` and
`self.bm [N,M,T]` behind exactly like the reference so that `remap=False` calls work -- in HBM, where the trellis
methods and `_estimate` read them; they become numpy arrays when somebody reads the attributes (see lnasr/hmm.py).

Covariances that are exactly diagonal take the fp64-MFMA kernel automatically; anything else
goes through the general (LU-inverse) kernel.  `si` may also be handed over already as the
diagonal `[N,M,D]` (extension).

Batch extras (not in the reference): `upload()`, `decode_batch`, `calc_prob_batch`,
`accumulate` / `apply_stats` (multi-utterance Baum-Welch, statistics all-reducible over RCCL).
"""

import numpy as np

from . import _ffi
from .hmm import HMM, _SERIAL, _save, _load, _log_normalised, _log_uniform, _report_iteration


class GMMHMM(HMM):

    def __init__(self, n: int = 1, m: int = 1, d: int = 1, A=None, pi=None, w=None, mu=None, si=None,
                 precision=np.double, *, ctx=None):
        super().__init__(n, m, A, None, pi, precision, ctx=ctx)
        self.d = d
        self.w = w
        self.mu = mu
        self.si = si
        self.min_std = 0.01

    # ------------------------------------------------------------------ plumbing
    # A, pi, w, mu, si are numpy arrays as in the reference -- except right after an on-device M-step (`train`), which
    # leaves the NEW parameters in HBM (25 MB of Sigma at 128 states x 16 mixtures x 39 x 39 need not visit the host between
    # two iterations).  The attributes are properties over the same mechanism as `b` / `bm` (lnasr/hmm.py): reading one copies
    # it to a numpy array, and from then on -- or when the caller assigns an array -- the host array is the parameter (it may
    # be edited in place) and goes through the fingerprinted upload again.
    _PARAMS = ("A", "pi", "w", "mu", "si")
    A = property(lambda self: self._host_get("A"), lambda self, value: self._host_set("A", value))
    pi = property(lambda self: self._host_get("pi"), lambda self, value: self._host_set("pi", value))
    w = property(lambda self: self._host_get("w"), lambda self, value: self._host_set("w", value))
    mu = property(lambda self: self._host_get("mu"), lambda self, value: self._host_set("mu", value))
    si = property(lambda self: self._host_get("si"), lambda self, value: self._host_set("si", value))

    def _upload_gmm(self, force_full=False):
        """All parameters -> device (per call on the reference API; unchanged parameters are recognised by their
        fingerprint and neither converted nor sent again, see _ffi.Context.upload_gmmhmm).

        Inside decode() / calc_prob() (`_speculating`) on a context whose model this object uploaded last, the check
        runs in the BACKGROUND (lnasr_hash128_begin) while the call's kernels are enqueued on the model the device
        holds; `HMM._confirmed` collects it before the answer is handed back, and a mismatch repeats the call the checked
        way.  The 0.2 ms fingerprint of a 25 MB model then overlaps the 0.8 ms the device needs anyway.

        Parameters an on-device M-step left in HBM are installed device to device, unchecked: nobody outside this object
        has seen them (reading an attribute turns it into a host array and sends the object down the checked road)."""
        if not force_full and self._params_current():
            return
        ctx = self.ctx
        res = self.__dict__.get("_resident", {})
        if any(k in res for k in self._PARAMS):
            if all(k in res and res[k][0].ctx is ctx and res[k][0].ptr for k in self._PARAMS):
                token = ("dev", self._dev_serial, bool(force_full))
                if ctx._model_fp != token:
                    d = {k: res[k][0] for k in self._PARAMS}
                    ctx.upload_gmmhmm_device(d["A"], d["pi"], d["w"], d["mu"], d["si"], diag=(len(d["si"].shape) == 3),
                                             force_full=force_full, owner=id(self), token=token)
                self._replaced = False
                return
            for k in self._PARAMS:                    # some read, some not: all of them to the host, the checked road below
                getattr(self, k)
        si = np.asarray(self.si)
        # (not after this object itself replaced its parameters -- _updatemodel, reset, load, apply_stats: the speculative run
        # would be thrown away for certain)
        if not force_full and self.__dict__.get("_speculating") and not self.__dict__.get("_replaced") \
                and ctx.model_owner == id(self) and ctx._model_fp is not None:
            self._spec_job = ctx.gmm_fingerprint_job(self.A, self.pi, self.w, self.mu, si, diag=(si.ndim == 3))
            return
        if force_full and si.ndim == 3:
            full = np.zeros(si.shape + (si.shape[-1],))
            idx = np.arange(si.shape[-1])
            full[..., idx, idx] = si
            si = full
        ctx.upload_gmmhmm(self.A, self.pi, self.w, self.mu, si, diag=(si.ndim == 3 and not force_full),
                          force_full=force_full, owner=id(self))
        self._replaced = False

    def upload(self):
        """Upload once before a series of *_batch calls (parameters must not change in between)."""
        self._upload_gmm()
        return self

    def _upload(self):
        # trellis-only calls need just A / pi, but keeping the whole model resident lets
        # `_estimate` read the old w / mu (gmmhmm.py:98,138) without another transfer
        self._upload_gmm()

    # ------------------------------------------------------------------ emission
    def _map_b(self, obs: np.ndarray):
        """bm[j,m,:] = log N(obs; mu[j,m], si[j,m]);  b[j,:] = lse_m(w[j,m] + bm[j,m,:])  (gmmhmm.py:46-67)."""
        if isinstance(obs, _ffi.DeviceArray):                  # features already in HBM (Recognizer: MFCC -> decode)
            batch = obs.view((1,) + obs.shape)
        else:
            batch = np.asarray(obs, dtype=np.float64)[None]
        self._upload_gmm()
        b, bm = self.ctx.gmm_logprob(batch, want_bm=True, precision=self.precision, device_out=True,
                                     b_out=self._device_buffer("b"), bm_out=self._device_buffer("bm"))
        self._keep("b", b, lambda h: h[0].T)                            # [1,T,N] -> [N,T]
        self._keep("bm", bm, lambda h: h[0].transpose(1, 2, 0))         # [1,T,N,M] -> [N,M,T]
        self._model_fresh = True                                        # see HMM._remap_or_upload

    bm = property(lambda self: self._host_get("bm"), lambda self, value: self._host_set("bm", value),
                  doc="per-mixture log-densities [N,M,T] of the last _map_b (gmmhmm.py:58-65)")

    # ------------------------------------------------------------------ M-step
    def _estimate(self, obs: np.ndarray, alpha, beta, xi, gamma):
        """{'A','pi','w','mu','si'}  (gmmhmm.py:69-156).  Sigma is taken around the OLD mean and
        gets `min_std * I` added, as in the reference."""
        obs = np.asarray(obs, dtype=np.float64)
        dt = np.float32 if np.asarray(gamma).dtype == np.float32 else np.float64
        self._upload_gmm()
        res = self.__dict__.get("_resident", {})
        if "bm" in res and res["bm"][0].dtype == np.dtype(dt):
            bm = res["bm"][0]                                                           # [1,T,N,M] in HBM
        else:
            bm = np.ascontiguousarray(np.asarray(self.bm).transpose(2, 0, 1), dtype=dt)     # [T,N,M]
        est = self.ctx.estimate_gmm(obs, bm, np.asarray(alpha, dtype=dt), np.asarray(beta, dtype=dt),
                                    np.asarray(xi, dtype=dt), np.asarray(gamma, dtype=dt), self.min_std)
        return est

    # ------------------------------------------------------------------ training
    def _fused_train_ok(self):
        """The on-device E+M step computes exactly what `_map_b` .. `_estimate` compute (with one
        utterance the accumulated statistics reduce to the reference's M-step), without moving xi [T,N,N] and
        bm [N,M,T] through host memory.  It is only taken when none of those steps is overridden and the model is
        float64; anything else walks the reference's sequence of calls (HMM.train)."""
        return self.precision in (np.double, np.float64) and self._ours(*self._REF_STEPS)

    def train(self, obs, iters=1, eps=0.0001, verbose: bool = True):
        """Baum-Welch on one sequence, stopping when |delta loglik| < eps  (hmm.py:285-313 with the
        gmmhmm.py:69-156 M-step): same iterates, same printed line.

        On the fused route everything between the observations going up and the likelihoods coming back stays in HBM: the
        packed statistics, the M-step's new A, pi, w, mu, Sigma (installed as the next model device to device; they become
        numpy arrays when somebody reads the attributes) -- per iteration two numbers cross PCIe.  And while another
        iteration may follow, L(lambda_new | O) is taken from the E-step that iteration needs anyway (the reference computes
        it twice: `calc_prob` at the end of iteration k and `lse(alpha[-1])` at the start of k + 1 are the same forward pass
        over the same model, hmm.py:301-304); the last iteration scores with `calc_prob` as the reference does, and when the
        loop stops on `eps` after an E-step, `_map_b` leaves `self.b` / `self.bm` as that `calc_prob` would have."""
        if not self._fused_train_ok():
            return super().train(obs, iters, eps, verbose)
        if iters <= 0:
            return                                     # the reference's `for k in range(iters)` does nothing
        ctx = self.ctx
        obs = np.ascontiguousarray(obs, dtype=np.float64)
        d_obs = ctx.to_device(obs[None])
        try:
            self._train_fused_loop(obs, d_obs, iters, eps, verbose)
        finally:
            d_obs.free()                               # also when a kernel call raises or the user interrupts

    def _train_fused_loop(self, obs, d_obs, iters, eps, verbose):
        ctx = self.ctx
        N, M, D = self.n, self.m, self.d
        ll_at = N * N + 2 * N + N * M + N * M * D + N * M * D * D

        def e_step():
            self._upload_gmm()
            if ctx.model_is_diag():                    # the reference's M-step returns a FULL Sigma even then
                self._upload_gmm(force_full=True)
            st = self._reuse_buffer("_stats_dev", (ctx.bw_stats_len(),))
            ctx.bw_accumulate(d_obs, stats_out=st)
            return st, self.precision(st.rows_to_host(ll_at, 1, 1)[0, 0])

        stats = None
        for k in range(iters):
            if stats is None:
                stats, before = e_step()
            spare = self.__dict__.get("_spare", {})
            self._updatemodel(ctx.bw_mstep(stats, self.min_std, device_out=True, out={p: spare.get(p) for p in self._PARAMS}))
            if k + 1 < iters:
                stats, after = e_step()                # = the `before` of iteration k + 1
            else:
                stats, after = None, self.calc_prob(obs)
            if _report_iteration(k, before, after, verbose) < eps:
                break
            before = after
        if stats is not None:                          # stopped on eps right after an E-step: the state calc_prob leaves
            self._map_b(d_obs.view(obs.shape))

    def _updatemodel(self, model):
        self._replaced = True
        for k in ("A", "pi", "mu", "si", "w"):
            if isinstance(model[k], _ffi.DeviceArray):
                self._keep_param(k, model[k])
            else:
                setattr(self, k, model[k])
        self._dev_serial = next(_SERIAL)

    def reset(self, init_type):
        """'uniform': zero means, identity Sigma; 'random': A, pi, w from the global numpy RNG (in that order, float64
        whatever `precision`), mu ~ U(-0.3, 0.3), Sigma = I  (gmmhmm.py:166-188).  Other values: silent no-op."""
        n, m, d = self.n, self.m, self.d
        if init_type == "uniform":
            self.A, self.pi, self.w = (_log_uniform(shape, self.precision) for shape in ((n, n), (n,), (n, m)))
            self.mu = np.zeros((n, m, d), dtype=self.precision)
        elif init_type == "random":
            self.A, self.pi, self.w = (_log_normalised(np.random.random_sample(shape)) for shape in ((n, n), (n,), (n, m)))
            self.mu = (0.6 * np.random.random_sample((n, m, d)) - 0.3).astype(self.precision)
        else:
            return
        self._replaced = True
        self.si = np.broadcast_to(np.eye(d, dtype=self.precision), (n, m, d, d)).copy()

    def save(self, filename):
        """Datasets 'A','pi','w','mu','si'  (gmmhmm.py:190-198)."""
        _save(filename, {"A": self.A, "pi": self.pi, "w": self.w, "mu": self.mu, "si": self.si})

    def load(self, filename):
        """gmmhmm.py:200-208; n, m, d are inferred from mu's shape."""
        d = _load(filename, ("A", "pi", "w", "mu", "si"))
        self.A, self.pi, self.w, self.mu, self.si = d["A"], d["pi"], d["w"], d["mu"], d["si"]
        self.n, self.m, self.d = self.mu.shape
        self._replaced = True

    # ------------------------------------------------------------------ batch extras
    def decode_batch(self, obs_batch, lens=None, want_score=False):
        """obs [U,T,D] (numpy or DeviceArray) -> paths [U,T] uint32: emission + Viterbi +
        backtrace fused on the device; call `upload()` first."""
        return self.ctx.decode(obs_batch, want_score=want_score, lens=lens)

    def calc_prob_batch(self, obs_batch, lens=None):
        return self.ctx.calc_prob(obs_batch, lens=lens)

    def accumulate(self, obs_batch, lens=None):
        """Baum-Welch sufficient statistics of a batch (packed fp64 vector; sum over ranks
        with Context.allreduce_f64, then `apply_stats`).  A model built with precision=np.float32 takes the
        fp32 matrix-core pass."""
        return self.ctx.bw_accumulate(obs_batch, lens=lens, precision=self.precision)

    def train_batch(self, obs_batch, lens=None, iters=1, eps=0.0001, verbose: bool = True, comm=None):
        """Baum-Welch over a BATCH of utterances (the reference's `train`, hmm.py:285-313, is single
        sequence): per iteration the statistics of this rank's utterances (`accumulate`, under the model's
        `precision=`), their sum over ranks when `comm` is given (an object with `allreduce_sum_f64`, e.g.
        `lnasr.dist.RcclComm`: one RCCL all-reduce of the packed vector), the M-step every rank runs identically,
        and the reference's stopping rule on the summed log-likelihood with the reference's printed line.
        Returns the list of (L(lambda|O), L(lambda_new|O)) pairs."""
        def e_step():
            st = np.asarray(self.accumulate(obs_batch, lens=lens))
            if comm is not None:
                st = np.asarray(comm.allreduce_sum_f64(np.ascontiguousarray(st, dtype=np.float64)))
            return st

        hist = []
        self.upload()
        stats = e_step()
        n_ll = len(stats) - 3                               # [..., loglik_sum, n_utt, n_frames]
        for k in range(iters):
            before = float(stats[n_ll])
            self.apply_stats(stats)
            self.upload()
            # L(lambda_new | O) from the SAME arithmetic as L(lambda | O): while another iteration may follow it is
            # the likelihood the next E-step produces anyway (one statistics pass per iteration instead of a pass
            # plus a scoring pass); on the last iteration a scoring pass under the model's precision
            if k + 1 < iters:
                stats = e_step()
                after = float(stats[n_ll])
            else:
                ll = np.asarray(self.ctx.calc_prob(obs_batch, lens=lens, precision=self.precision), dtype=np.float64)
                new = np.array([ll.sum()])
                if comm is not None:
                    new = np.asarray(comm.allreduce_sum_f64(new))
                after = float(new[0])
            hist.append((before, after))
            if _report_iteration(k, before, after, verbose) < eps:
                break
        return hist

    def apply_stats(self, stats):
        """M-step from accumulated statistics; updates the parameters in place."""
        m = self.ctx.bw_mstep(stats, self.min_std)
        self._replaced = True
        self.A, self.pi, self.w, self.mu = m["A"], m["pi"], m["w"], m["mu"]
        self.si = m["var"] if "var" in m else m["si"]
        return m
