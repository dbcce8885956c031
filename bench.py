#!/usr/bin/env python3
"""Headline benchmark: frames/s of GMM-HMM Viterbi decode, N=128 states, M=16 mixtures, D=39
(BASELINE.json metric; workload = configs[3]: T=1000, 8192 utterances over 8 GPUs, i.e. 1024
utterances per GPU, weak scaling -- utterances shard with no data-path collective except the
final RCCL all-gather of the decoded state paths).

    python bench.py [--gpus N] [--steps K] [--warmup W]

With --gpus N > 1 and no launcher environment (WORLD_SIZE unset) this process touches no GPU: it
starts N fresh ranks of itself, one per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT in their environment), forwards their output and returns the largest exit code.
Started by somebody else's launcher (the driver uses an elastic launcher with --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 --master-port P in front of `bench.py --gpus N ...`)
the process is one rank and reads the same variables; whoever launched it, the environment
defaults a rank needs on this pool are set below, at import.  No ML framework anywhere in this
file or in the package: the ranks meet through lnasr/rendezvous.py (loopback TCP; the 128-byte
ncclUniqueId, the barriers and the max-over-ranks of the timing travel there).

A step = one pass of the hot path (fp64-MFMA emission -> Viterbi trellis -> backtrace, then for
N > 1 the RCCL gather of paths) over this rank's 1024 utterances, with the observations already
resident in HBM.  Rank 0 prints ONE JSON line.  The compute path is liblnasr_hip.so through ctypes
and the collectives are RCCL inside that library.

Before anything is timed EVERY rank decodes the first N_CHECK utterances of its own shard with the
CPU oracle and compares the paths bit for bit; a mismatch, a failed communicator or a failed
rehearsal gather means NOTHING is timed: rank 0 prints the line with "value": null and the reason,
and every rank exits non-zero.

Besides the headline the line carries (inside `roofline`: `traffic` with its provenance, `shader_clock` = the clock the chip is
left at behind the emission kernel, behind an fp64 kernel and idle)
  end_to_end   host numpy observations in -> host paths out (the reference's decode(), hmm.py:276-283,
               takes host arrays): page-locked buffers through the chunked upload pipeline, and the
               same from pageable memory through the pinned staging ring; PCIe-inclusive, never `value`;
  baum_welch   BASELINE configs[4]: one statistics pass per rank (N=256, M=32, T=2000, 64 utterances:
               emission + forward/backward + accumulation) and the RCCL all-reduce of the packed
               sufficient statistics (the second collective of the path), timed separately; `roofline`: the
               pass's algorithmic FLOP over its time against the fp64 / fp32 peak + its three
               dominant kernels;
  configs3_one_gpu  (one rank only) all 8192 utterances of BASELINE configs[3] on this one GPU, same resident step;
  reference_api  (one rank only) the reference's own method names on ONE utterance through the mirror
               classes -- GMMHMM.decode / calc_prob / _viterbi / _forward / train(iters=1) at the metric
               shape with si as [N,M,D,D]: wall, device and host ms per call (tools/refapi.py).

    --dry-launch    CPU rehearsal of the launch path (tests): the ranks rendezvous, shard a toy batch,
                    gather toy paths through the CPU transport (lnasr.dist.TcpComm) and rank 0 prints a
                    {"dry_launch": true, "n_ranks": N, ...} line.  No GPU call, no timing.
"""
import os

# Defaults every rank needs, whoever started it (bench.py's own launcher, the driver's elastic launcher, a
# test): set at import, before anything reads them.
_ENV_DEFAULTS = {
    "HSA_ENABLE_IPC_MODE_LEGACY": "0",      # dmabuf IPC: RCCL across processes fails without it on this pool
    "GLOO_SOCKET_IFNAME": "lo",             # one node: never depend on the hostname resolving (a foreign launcher's gloo) ...
    "NCCL_SOCKET_IFNAME": "lo",             # ... nor on which interface RCCL's bootstrap would pick
    "MASTER_ADDR": "127.0.0.1",
    "OMP_NUM_THREADS": "1", "OPENBLAS_NUM_THREADS": "1", "MKL_NUM_THREADS": "1",   # CPU baseline: 1 BLAS thread / worker
}
for _k, _v in _ENV_DEFAULTS.items():
    os.environ.setdefault(_k, _v)

import argparse
import importlib.util
import json
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "ln-asr_amd"))

N_STATES, N_MIX, N_DIM, T_FRAMES, UTT_PER_GPU = 128, 16, 39, 1000, 1024
FLOP_EMIT = 3 * N_STATES * N_MIX * N_DIM            # 239,616 per frame
FLOP_TRELLIS = 2 * N_STATES * N_STATES              # 32,768 per frame
FLOP_DECODE = FLOP_EMIT + FLOP_TRELLIS              # 272,384 per frame
BYTES_DECODE = 8 * N_DIM + 4                        # 316 B per frame (obs in, path out)
PEAK_FP64_TFLOPS = 78.6                             # MI355X fp64 matrix = vector peak (MI355X_MICROARCH.md)
PEAK_FP32_TFLOPS = 157.3                            # fp32 vector / matrix peak (same guide)
PEAK_HBM_GBS = 8000.0
PEAK_I8_TOPS = 5000.0                               # dense int8 matrix peak: 2 x the bf16 rate per clock (same guide)
METRIC = "frames/sec Viterbi decode, 128-state 16-mix GMM-HMM, D=39 MFCC"
N_CHECK = 4                                         # utterances per rank compared with the oracle before timing
BW_SHAPE = dict(N=256, M=32, D=39, T=2000, U=64)    # BASELINE configs[4], per GPU
BW_STEPS = 5


_T0 = time.time()


def note(msg):
    """Progress line on stderr (rank 0 says where the run is; a hung stage is then visible in the log)."""
    if int(os.environ.get("RANK", "0")) == 0:
        sys.stderr.write("[bench +%.1fs] %s\n" % (time.time() - _T0, msg))
        sys.stderr.flush()


def _load_oracle():
    spec = importlib.util.spec_from_file_location("lnasr_oracle", os.path.join(REPO, "oracle", "lnasr_oracle.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ---- launcher: --gpus N > 1 without a launcher environment ----------------------------------------------
def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args, argv):
    """Start N fresh ranks (one per GPU) from a parent that has made no GPU call, wait, return the largest exit code.
    A rank that fails takes the others with it after a grace period (they would wait for it in the rendezvous)."""
    base = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_PORT"):
        base.pop(k, None)
    port = _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    sys.stderr.write("[bench] started %d ranks of %s %s (pids %s), rendezvous key 127.0.0.1:%d\n"
                     % (args.gpus, os.path.basename(__file__), " ".join(argv), [p.pid for p in procs], port))
    sys.stderr.flush()
    codes, failed_at = [None] * len(procs), None
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
                if codes[i] not in (None, 0) and failed_at is None:
                    failed_at = time.time()
        if failed_at is not None and time.time() - failed_at > 30.0:
            for i, p in enumerate(procs):                          # exactly the processes started above
                if codes[i] is None:
                    p.kill()
                    codes[i] = p.wait()
        time.sleep(0.05)
    return max(abs(c) for c in codes)


# ---- CPU baseline (rank 0, N=1 only): the oracle timed on the host cores ------------------------------
_W = {}


def _cpu_worker(args):
    seed, n_utt = args
    from lnasr import synth
    if "o" not in _W:
        _W["o"] = _load_oracle()
        mdl = synth.synth_model(N_STATES, N_MIX, N_DIM, seed=1)
        _W["m"] = (mdl["A"], mdl["pi"], _W["o"].prep_diag_fused(mdl["w"], mdl["mu"], mdl["var"]))
    o = _W["o"]
    A, pi, Wc = _W["m"]
    obs = synth.synth_obs(n_utt, T_FRAMES, N_DIM, seed=seed)
    t0 = time.perf_counter()
    chk = 0
    for u in range(n_utt):
        chk += int(o.decode_diag_fused(obs[u], A, pi, Wc, N_STATES, N_MIX).sum())
    return time.perf_counter() - t0, chk


def _lscpu():
    """CPU model / sockets / cores of the box (always state them)."""
    want = {"Model name": "model", "Socket(s)": "sockets", "Core(s) per socket": "cores_per_socket",
            "Thread(s) per core": "threads_per_core", "CPU(s)": "logical_cpus"}
    out = {}
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=20).stdout
        for line in txt.splitlines():
            k, _, v = line.partition(":")
            if k.strip() in want and want[k.strip()] not in out:
                out[want[k.strip()]] = v.strip()
    except Exception:                                                  # noqa: BLE001
        pass
    return out


def _cpu_quota():
    """CPUs this process may actually burn (cgroup quota), which sched_getaffinity does not show: a 1-GPU box sees all
    of the host's logical CPUs but owns a share of them."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as fp:                         # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = fp.read().split()[:2]
            if q != "max":
                return max(1, int(int(q) / int(per)))
    except Exception:                                                      # noqa: BLE001
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
            q, per = int(fq.read()), int(fp.read())
            if q > 0:
                return max(1, q // per)
    except Exception:                                                      # noqa: BLE001
        pass
    return None


def oracle_paths(check_obs):
    """The checker's answer: the oracle's Viterbi paths of a few utterances of the bench's own input."""
    from lnasr import synth
    o = _load_oracle()
    mdl = synth.synth_model(N_STATES, N_MIX, N_DIM, seed=1)
    Wc = o.prep_diag_fused(mdl["w"], mdl["mu"], mdl["var"])
    return np.stack([o.decode_diag_fused(x, mdl["A"], mdl["pi"], Wc, N_STATES, N_MIX) for x in check_obs])


def cpu_baseline():
    """Line B of BASELINE.md: best vectorised numpy form, one process per host core, one BLAS
    thread each, disjoint utterances; plus line A (reference-shaped loops, 1 core) as context."""
    import multiprocessing as mp
    avail = len(os.sched_getaffinity(0))
    quota = _cpu_quota()
    if quota is not None:
        avail = min(avail, quota)
    cpu = _lscpu()
    cpu["usable_by_this_process"] = avail
    cpu["cgroup_cpu_quota"] = quota
    try:
        socket_cores = int(cpu.get("cores_per_socket", "0"))
    except ValueError:
        socket_cores = 0
    # BASELINE.md section 3 line B names ONE SOCKET's cores.  A 1-GPU box owns 16 of the host's logical CPUs (pool
    # guidance; the affinity mask shows them all, a cgroup quota -- when there is one -- shows the share): when this
    # process may use a whole socket's worth AND LNASR_CPU_SOCKET=1 asks for it, that many workers are timed; otherwise 16
    # (or what is usable), and the single-socket figure is given as a labelled estimate.  (Round 4: 64 workers on a
    # box whose mask showed 128 CPUs ran at 5.5 k frames/s per worker against 27 k with 16 -- they shared the box's
    # share of the host -- so oversubscribing by default would UNDER-state the baseline.)  LNASR_CPU_WORKERS overrides.
    if "LNASR_CPU_WORKERS" in os.environ:
        cores = max(1, min(avail, int(os.environ["LNASR_CPU_WORKERS"])))
    elif socket_cores and avail >= socket_cores and os.environ.get("LNASR_CPU_SOCKET"):
        cores = socket_cores
    else:
        cores = max(1, min(avail, 16))
    per = int(os.environ.get("LNASR_CPU_UTTS", "400" if cores <= 16 else "250"))     # ~10-15 s per worker at ~27 k frames/s per core
    ctx = mp.get_context("fork")
    with ctx.Pool(cores) as pool:
        pool.map(_cpu_worker, [(5000 + i, 1) for i in range(cores)])          # warm caches / imports
        t0 = time.perf_counter()
        pool.map(_cpu_worker, [(6000 + i, per) for i in range(cores)])
        wall = time.perf_counter() - t0
    frames = cores * per * T_FRAMES
    per_core = frames / wall / cores
    out = {"value": frames / wall, "unit": "frames/s", "cores": cores, "kind": "port",
           "per_core": per_core,
           "single_socket_estimate": (socket_cores * per_core) if socket_cores else None,
           "single_socket_estimate_note": (
               "measured: the workers timed ARE one socket's cores" if socket_cores and cores == socket_cores else
               "ESTIMATE = cores_per_socket (%s) x per_core, linear in cores (optimistic: memory bandwidth and turbo do "
               "not scale that way); only %d of the socket's cores were timed -- gpu_over_cpu below is against those %d, "
               "NOT the single-socket ratio the north-star target names (gpu_over_single_socket_estimate is)"
               % (socket_cores or "?", cores, cores)),
           "sample": "%d utterances x %d frames (same synthetic model/shape), oracle decode_diag_fused "
                     "(numpy %s GEMM emission + vectorised Viterbi), %d worker processes x 1 BLAS thread, %.1f s wall; "
                     "host: %s, %s socket(s) x %s cores x %s threads, %s logical CPUs (%d usable by this process)"
                     % (cores * per, T_FRAMES, np.__version__, cores, wall, cpu.get("model", "?"), cpu.get("sockets", "?"),
                        cpu.get("cores_per_socket", "?"), cpu.get("threads_per_core", "?"), cpu.get("logical_cpus", "?"),
                        avail),
           "host": cpu}
    # line A: reference-shaped loop structure, single core, short prefix
    from lnasr import synth
    o = _load_oracle()
    mdl = synth.synth_model(N_STATES, N_MIX, N_DIM, seed=1)
    obs = synth.synth_obs(1, 200, N_DIM, seed=1000)[0]
    t0 = time.perf_counter()
    o.decode_refshaped(obs, mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["si"])
    ref_shaped = 200 / (time.perf_counter() - t0)
    return out, {"value": ref_shaped, "unit": "frames/s", "cores": 1,
                 "sample": "1 utterance x 200 frames, per-(state,mixture) density calls + per-(frame,state) "
                           "trellis updates (the reference's loop structure)"}


def _pmc_traffic(kernel):
    """(HBM bytes per launch, where the figure comes from) of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/pmc_traffic.json, corrected as MI355X_MICROARCH.md section HBM prescribes).  The counters are collected in
    their own profiler passes, not in this run: the entry names the profile (tag, date) and carries a hash of the source
    the counted kernel was built from (tools/prof_summary.py) -- a kernel whose source has changed since gets NO figure."""
    import hashlib
    p = os.path.join(REPO, "profiles", "pmc_traffic.json")
    try:
        with open(p) as fp:
            ent = json.load(fp).get(kernel)
        if not ent:
            return None, "no entry for %s in profiles/pmc_traffic.json" % kernel
        if "source_sha16" not in ent:
            return None, "profiles/pmc_traffic.json entry predates source tracking: not used"
        h = hashlib.sha256()
        for f in (os.path.basename(ent["source"]), "devmath.h"):
            with open(os.path.join(REPO, "ln-asr_amd", "csrc", f), "rb") as fp:
                h.update(fp.read())
        if h.hexdigest()[:16] != ent["source_sha16"]:
            return None, "STALE: %s changed after profile %s (%s); re-run tools/profile_round.sh" % (
                ent["source"], ent.get("tag"), ent.get("date"))
        return ent.get("hbm_bytes_per_launch"), "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, profile %s (%s), %s unchanged since" % (
            ent.get("tag"), ent.get("date"), ent["source"])
    except Exception as e:                                             # noqa: BLE001
        return None, "unreadable: %s" % e


# ---- --dry-launch: the launch path on CPU ----------------------------------------------------------------
def dry_launch(rank, world):
    from lnasr import dist as ldist
    from lnasr import rendezvous
    grp = rendezvous.TcpGroup(rank, world)
    try:
        U, T = 4 * world + 1, 8                                           # ragged shards
        lo, hi = ldist.shard_range(U, rank, world)
        truth = (np.arange(U * T, dtype=np.uint32).reshape(U, T) * 7) % N_STATES
        comm = ldist.TcpComm(grp)
        allp = ldist.gather_paths(comm, truth[lo:hi], U, T)
        ok = bool(np.array_equal(allp, truth))
        seen = grp.allgather([rank, int(os.environ.get("LOCAL_RANK", "-1")), os.getpid(),
                              {k: os.environ.get(k) for k in _ENV_DEFAULTS}])
        if rank == 0:
            print(json.dumps({"dry_launch": True, "n_ranks": world, "ranks": sorted(s[0] for s in seen),
                              "local_ranks": sorted(s[1] for s in seen), "distinct_pids": len({s[2] for s in seen}),
                              "gather_ok": ok, "transport": "loopback TCP (CPU rehearsal, nothing timed)",
                              "env": [s[3] for s in sorted(seen, key=lambda s: s[0])]}))
            sys.stdout.flush()
        grp.barrier()
    finally:
        grp.close()
    return 0 if ok else 1


# ---- the rank ---------------------------------------------------------------------------------------------
class Ranks:
    """Rendezvous / barrier / reductions between ranks (lnasr/rendezvous.py).  One rank: no-ops, no socket."""

    def __init__(self, rank, world, use_comm):
        from lnasr import rendezvous
        self.rank, self.world = rank, world
        self.grp = rendezvous.TcpGroup(rank, world) if use_comm else None

    def barrier(self):
        if self.grp is not None:
            self.grp.barrier()

    def max(self, x):
        return float(x) if self.grp is None else self.grp.max(x)

    def gather(self, obj):
        return [obj] if self.grp is None else self.grp.allgather(obj)

    def bcast(self, obj):
        return obj if self.grp is None else self.grp.bcast(obj)

    def close(self):
        if self.grp is not None:
            self.grp.close()


def fail_line(rk, why):
    """Nothing was timed: say so on rank 0 and leave non-zero on every rank (never through a re-exec; a wedged
    helper thread may still sit inside the library, hence os._exit)."""
    if rk.rank == 0:
        print(json.dumps({"metric": METRIC, "value": None, "unit": "frames/s", "n_gpus": 0, "steps": 0, "warmup": 0,
                          "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "f64", "data": "synthetic",
                          "config": {"workload": "not run", "gather": "FAILED, nothing timed: %s" % why,
                                     "parity": "FAILED, nothing timed: %s" % why}}))
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(1)


def init_comm(ctx, rk, d_path, U, T):
    """RCCL communicator + one rehearsal gather under a watchdog.  Returns (gather buffer, ranks RCCL counts)."""
    import threading
    from lnasr import _ffi
    state = {}

    def _work():
        try:
            ctx.comm_init(state["id"], rk.rank, rk.world)
            buf = ctx.empty((rk.world, U, T), np.uint32)
            ctx.allgather_paths(d_path, N_STATES, out=buf)
            ctx.sync()
            state["info"] = ctx.comm_info()
            state["buf"] = buf
        except Exception as e:                                   # noqa: BLE001
            state["err"] = e

    err = None
    try:
        state["id"] = rk.bcast(_ffi.comm_unique_id() if rk.rank == 0 else None)
        th = threading.Thread(target=_work, daemon=True)
        th.start()
        th.join(float(os.environ.get("LNASR_COMM_TIMEOUT", "180")))
        if th.is_alive():
            raise RuntimeError("RCCL bootstrap / first all-gather did not finish in time")
        if "err" in state:
            raise state["err"]
        if state["info"] != (rk.rank, rk.world):
            raise RuntimeError("RCCL reports rank/size %r, the launcher %r" % (state["info"], (rk.rank, rk.world)))
    except Exception as e:                                       # noqa: BLE001
        err = "%s: %s" % (type(e).__name__, e)
        sys.stderr.write("[rank %d] RCCL init failed: %s\n" % (rk.rank, err))
    errs = rk.gather(err)
    if any(errs):
        fail_line(rk, "; ".join("rank %d: %s" % (r, e) for r, e in enumerate(errs) if e))
    return state["buf"], state["info"][1]


def timed(rk, ctx, fn, steps, warmup):
    """W untimed + exactly K timed calls of fn, bracketed by barrier + device sync on both sides; max over ranks (s)."""
    for _ in range(warmup):
        fn()
    ctx.device_sync()
    rk.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    ctx.device_sync()
    dt = time.perf_counter() - t0
    rk.barrier()
    return rk.max(dt)


def end_to_end(rk, ctx, obs, want_paths, compute_ms, steps):
    """Host observations in -> host paths out (the reference's decode() signature), PCIe inclusive."""
    U, T = obs.shape[:2]
    out = {}
    pin_obs = ctx.pinned_copy(obs)
    pin_path = ctx.pinned_empty((U, T), np.uint32)
    d_tmp = ctx.empty(obs.shape, np.float64)
    from lnasr import _ffi

    def h2d():
        _ffi.check(_ffi.lib().lnasr_memcpy_h2d(ctx.handle, d_tmp.ptr, pin_obs.ctypes.data, obs.nbytes))
    h2d_s = timed(rk, ctx, h2d, 5, 1) / 5
    d_tmp.free()

    def run_pinned():
        ctx.decode(pin_obs, path_out=pin_path)
    t_pin = timed(rk, ctx, run_pinned, steps, 2) / steps
    ok_pin = bool(np.array_equal(pin_path, want_paths))
    page_path = np.empty((U, T), np.uint32)

    def run_pageable():
        ctx.decode(obs, path_out=page_path)
    t_page = timed(rk, ctx, run_pageable, max(3, steps // 2), 2) / max(3, steps // 2)
    ok_page = bool(np.array_equal(page_path, want_paths))
    frames = rk.world * U * T
    floor_ms = max(h2d_s * 1e3, compute_ms)
    out = {"value": frames / t_pin, "unit": "frames/s", "ms_per_step": t_pin * 1e3,
           "h2d_GBps": obs.nbytes / h2d_s / 1e9, "h2d_ms": h2d_s * 1e3, "compute_ms": compute_ms,
           "over_max_h2d_compute": t_pin * 1e3 / floor_ms,
           "what": "numpy float64 obs [%d,%d,%d] in page-locked host memory (lnasr_host_alloc) -> uint32 paths in "
                   "page-locked host memory; upload in chunks on a side stream, emission starts on the first chunk; "
                   "h2d_GBps = the same buffer copied alone" % obs.shape,
           "paths_identical_to_resident_run": ok_pin,
           "pageable": {"value": frames / t_page, "ms_per_step": t_page * 1e3,
                        "what": "plain (pageable) numpy arrays in and out: host threads fill the context's pinned staging "
                                "ring, DMA from there", "paths_identical_to_resident_run": ok_page}}
    return out, ok_pin and ok_page


def baum_welch(rk, ctx, use_rccl):
    """BASELINE configs[4] per rank: statistics pass (fp64 and precision=float32) + RCCL all-reduce of the packed vector."""
    from lnasr import synth
    s = BW_SHAPE
    mdl = synth.synth_model(s["N"], s["M"], s["D"], seed=5)
    obs = synth.synth_obs(s["U"], s["T"], s["D"], seed=2000 + rk.rank)
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
    d_obs = ctx.to_device(obs)
    n = ctx.bw_stats_len()
    d_stats = ctx.empty((n,), np.float64)
    out = {"workload": "Baum-Welch statistics pass, N=%(N)d M=%(M)d D=%(D)d T=%(T)d, %(U)d utterances per GPU "
                       "(BASELINE configs[4]): emission + forward/backward + accumulation; then the RCCL sum "
                       "all-reduce of the packed sufficient statistics" % s,
           "steps": BW_STEPS, "allreduce_bytes": n * 8}
    frames = s["U"] * s["T"]
    for tag, prec in (("f64", np.float64), ("f32", np.float32)):
        t = timed(rk, ctx, lambda: ctx.bw_accumulate(d_obs, stats_out=d_stats, precision=prec), BW_STEPS, 1) / BW_STEPS
        out["bw_pass_ms" if tag == "f64" else "bw_pass_f32_ms"] = t * 1e3
    ctx.bw_accumulate(d_obs, stats_out=d_stats, precision=np.float64)
    ctx.sync()
    mine = d_stats.to_host()
    # size-independent checks of this rank's statistics: sum_j gamma = frames - U (the last frame of every utterance
    # carries no transition), sum of the mixture posteriors = frames, n_utt / n_frames as counted
    N, M = s["N"], s["M"]
    g = mine[N * N:N * N + N].sum()
    mix = mine[N * N + 2 * N:N * N + 2 * N + N * M].sum()
    ok = abs(g - (frames - s["U"])) < 1e-6 * frames and abs(mix - frames) < 1e-6 * frames
    ok = ok and mine[-2] == s["U"] and mine[-1] == frames
    out["allreduce_ms"] = None
    if use_rccl:
        # the all-reduce alone, in place on the device-resident vector (re-filled before each call so that the
        # values stay finite whatever the step count)
        ts = []
        for _ in range(BW_STEPS + 1):
            ctx.bw_accumulate(d_obs, stats_out=d_stats, precision=np.float64)
            ctx.device_sync()
            rk.barrier()
            t0 = time.perf_counter()
            ctx.allreduce_f64(d_stats)
            ctx.sync()
            ts.append(time.perf_counter() - t0)
        out["allreduce_ms"] = rk.max(float(np.median(ts[1:]))) * 1e3
        total = d_stats.to_host()
        sums = rk.gather(float(mine[-3]))                               # every rank's own log-likelihood sum
        ok = ok and abs(total[-3] - sum(sums)) <= 1e-9 * abs(sum(sums))
        ok = ok and total[-2] == s["U"] * rk.world and total[-1] == frames * rk.world
        out["allreduce_GBps_algorithmic"] = n * 8 / (out["allreduce_ms"] * 1e-3) / 1e9
        # every rank then runs the identical M-step on the summed statistics
        m = ctx.bw_mstep(total)
        ok = ok and bool(np.allclose(np.exp(m["A"]).sum(axis=1), 1.0, atol=1e-9))
    out["frames_per_s"] = rk.world * frames / ((out["bw_pass_ms"] + (out["allreduce_ms"] or 0.0)) * 1e-3)
    # roofline of the pass ("BW statistics (diag)"): emission 3 N M D + forward and backward 2 x 3 N^2 + xi N^2 +
    # the mixture moments 4 N M D per frame -- algorithmic FLOP over the driver-timed pass, against the fp64 peak and, for the
    # precision=float32 pass, the fp32 one; beside it the three kernels that take most of the pass (HIP events, one more pass)
    flop = 3 * N * M * s["D"] + 2 * 3 * N * N + N * N + 4 * N * M * s["D"]
    roof = {"bound": "mfma", "algorithmic_flop_per_frame": flop, "frames_per_pass": frames, "unit": "TFLOP/s",
            "what": "3NMD (emission) + 2*3N^2 (forward, backward) + N^2 (xi) + 4NMD (moments) per frame, N=%d M=%d D=%d" % (N, M, s["D"])}
    for tag, prec, peak, key in (("f64", np.float64, PEAK_FP64_TFLOPS, "bw_pass_ms"), ("f32", np.float32, PEAK_FP32_TFLOPS, "bw_pass_f32_ms")):
        ach = flop * frames / (out[key] * 1e-3) / 1e12
        ctx.device_sync()
        ctx.prof_enable(True)
        ctx.prof_reset()
        ctx.bw_accumulate(d_obs, stats_out=d_stats, precision=prec)
        ctx.device_sync()
        ctx.prof_enable(False)
        top = sorted(ctx.prof_report().items(), key=lambda kv: -kv[1][1])[:3]
        roof[tag] = {"achieved": ach, "peak": peak, "frac": ach / peak, "pass_ms": out[key],
                     "dominant_kernels_ms": {k: m for k, (n, m) in top}}
    out["roofline"] = roof
    out["checks"] = "sum(gamma)=frames-U, sum(mix)=frames, counts%s: %s" % (
        ", all-reduced loglik = sum of the ranks', normalised M-step" if use_rccl else "", "ok" if ok else "FAILED")
    oks = rk.gather(bool(ok))
    d_obs.free()
    d_stats.free()
    return out, all(oks)


def full_config3(ctx, mdl, first_shard_paths):
    """BASELINE configs[3] in full on ONE GPU: all 8192 utterances (the eight shards' seeds, 2.6 GB of observations, 8.4 GB
    each of b and v) through the same resident step, on the driver's clock.  Checks: shard 0 is the headline's shard, so its
    paths must be the headline's bit for bit; every state index in range."""
    from lnasr import synth, _ffi
    shards, U, T = 8, UTT_PER_GPU, T_FRAMES
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
    d_obs = ctx.empty((shards * U, T, N_DIM), np.float64)
    shard_bytes = U * T * N_DIM * 8
    for k in range(shards):                                              # shard by shard: 320 MB of host memory at a time
        x = synth.synth_obs(U, T, N_DIM, seed=1000 + k)
        _ffi.check(_ffi.lib().lnasr_memcpy_h2d(ctx.handle, d_obs.ptr + k * shard_bytes, x.ctypes.data, shard_bytes))
    d_path = ctx.empty((shards * U, T), np.uint32)

    class _One:                                                          # timed() wants the rank helpers
        def barrier(self): pass
        def max(self, x): return float(x)
    steps = 3
    t = timed(_One(), ctx, lambda: ctx.decode(d_obs, path_out=d_path), steps, 1) / steps
    p = d_path.to_host()
    out = {"workload": "all %d utterances of BASELINE configs[3] on one GPU, same resident step" % (shards * U),
           "utterances": shards * U, "steps": steps, "ms_per_step": t * 1e3, "value": shards * U * T / t, "unit": "frames/s",
           "first_shard_paths_identical_to_headline": bool(np.array_equal(p[:U], first_shard_paths)),
           "all_states_in_range": bool(p.max() < N_STATES)}
    d_obs.free()
    d_path.free()
    return out, out["first_shard_paths_identical_to_headline"] and out["all_states_in_range"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--utts", type=int, default=UTT_PER_GPU, help="utterances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the end_to_end and baum_welch sections")
    ap.add_argument("--dry-launch", action="store_true", help="CPU rehearsal of the N-rank launch path (no GPU)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # the very first thing, before any import that touches the GPU: N fresh child ranks
        return launch_ranks(args, sys.argv[1:])

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("[rank %d] --gpus %d but the launcher started %d ranks; reporting the ranks that run\n"
                         % (rank, args.gpus, world))
    if args.dry_launch:
        return dry_launch(rank, world)
    U, T = args.utts, T_FRAMES

    from lnasr import synth
    obs = synth.synth_obs(U, T, N_DIM, seed=1000 + rank)            # this rank's shard

    # everything that forks (CPU-baseline workers) or is pure CPU comes first: before this process touches the GPU
    cpu, cpu_ref = (None, None)
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        note("CPU baseline (oracle on the host cores, ~20 s)")
        cpu, cpu_ref = cpu_baseline()
    note("oracle paths of the first %d utterances" % min(N_CHECK, U))
    n_check = min(N_CHECK, U)
    want_head = oracle_paths(obs[:n_check])                         # every rank checks its own shard
    if os.environ.get("LNASR_BENCH_TEST_CORRUPT_ORACLE"):           # tests only: prove that a mismatch times nothing
        want_head[0, 0] ^= 1

    # LNASR_BENCH_FORCE_COMM=1: take the N > 1 code path (gloo group, RCCL communicator under the watchdog, gather in
    # the timed step, all-reduce of the statistics) with however many ranks there are -- on a one-GPU box a one-rank
    # rehearsal of everything but the links
    use_comm = world > 1 or bool(os.environ.get("LNASR_BENCH_FORCE_COMM"))
    rk = Ranks(rank, world, use_comm)

    from lnasr import _ffi
    # one GPU per rank: LOCAL_RANK (a launcher that isolates one GPU per rank leaves one visible device: that one).
    # LNASR_BENCH_DEVICE pins every rank to one device -- only the failure-path test uses it (RCCL refuses two ranks
    # on one GPU)
    if "LNASR_BENCH_DEVICE" in os.environ:
        dev = int(os.environ["LNASR_BENCH_DEVICE"])
    else:
        dev = 0 if _ffi.device_count() == 1 else local_rank
    note("context on device %d" % dev)
    ctx = _ffi.Context(dev)

    mdl = synth.synth_model(N_STATES, N_MIX, N_DIM, seed=1)
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
    model_range = ctx.model_range()                                 # (rho, integer bank admitted) of the metric's model
    d_obs = ctx.to_device(obs)
    d_path = ctx.empty((U, T), np.uint32)
    d_score = ctx.empty((U,), np.float64)
    d_all = None
    n_ranks_rccl = 1
    gather = "n/a (single GPU)"
    if use_comm:
        note("RCCL communicator + rehearsal gather")
        d_all, n_ranks_rccl = init_comm(ctx, rk, d_path, U, T)
        gather = ("rccl all-gather of the state paths over %d ranks (ncclCommCount; u8 on the wire, u32 in memory), "
                  "%d B per GPU per step" % (n_ranks_rccl, U * T))

    def step():
        ctx.decode(d_obs, path_out=d_path, score_out=d_score)
        if d_all is not None:
            ctx.allgather_paths(d_path, N_STATES, out=d_all)

    # the checker's answer first: a fast kernel with different paths is not measured -- on ANY rank
    note("parity check, then %d warm-up + %d timed steps" % (args.warmup, args.steps))
    step()
    ctx.sync()
    resident_paths = d_path.to_host()
    bad = None if np.array_equal(resident_paths[:n_check], want_head) else \
        "rank %d: GPU paths differ from the oracle's on the bench's own input" % rank
    bads = [b for b in rk.gather(bad) if b]
    if bads:
        sys.stderr.write("; ".join(bads) + "; nothing timed\n")
        fail_line(rk, "; ".join(bads))
    parity = ("paths of the first %d utterances of every rank's shard (%d rank%s) bit-identical to the oracle's "
              "before timing" % (n_check, world, "" if world == 1 else "s"))

    for _ in range(args.warmup):
        step()
    ctx.device_sync()
    rk.barrier()
    ctx.prof_enable(True)
    ctx.prof_reset()
    ctx.device_sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.device_sync()
    elapsed = time.perf_counter() - t0
    rk.barrier()
    ctx.prof_enable(False)
    prof = ctx.prof_report()
    elapsed = rk.max(elapsed)

    # sanity: the decoded paths are valid state indices and every rank holds every shard
    p = d_path.to_host()
    rc = 0
    if p.max() >= N_STATES or not np.array_equal(p, resident_paths):
        sys.stderr.write("[rank %d] decoded paths out of range or not reproducible\n" % rank)
        rc = 1
    if d_all is not None:
        allp = d_all.to_host()
        if not np.array_equal(allp[rank], p):
            sys.stderr.write("[rank %d] gathered paths do not match the local shard\n" % rank)
            rc = 1
        sums = rk.gather(int(p.astype(np.uint64).sum()))
        if [int(allp[r].astype(np.uint64).sum()) for r in range(world)] != sums:
            sys.stderr.write("[rank %d] gathered shards differ from what their owners decoded\n" % rank)
            rc = 1

    step_s = elapsed / args.steps
    # the shader clock the chip is left at (lnasr_debug_clock: a 20 us one-wave probe directly behind the kernel): the integer
    # matrix kernel is power-limited, and what follows it -- the Viterbi kernel, 1.3 ms -- inherits its clock
    clocks = None
    try:
        b_tmp = ctx.empty((U, T, N_STATES), np.float64)
        ctx.gmm_logprob(d_obs, b_out=b_tmp)
        ck_emit = ctx.debug_clock()
        ctx.forward(b_tmp, want_alpha=False, want_loglik=True)
        ck_f64 = ctx.debug_clock()
        ctx.device_sync()
        time.sleep(0.05)
        clocks = {"behind_the_emission_kernel_ghz": ck_emit, "behind_an_fp64_vector_kernel_ghz": ck_f64,
                  "after_50_ms_idle_ghz": ctx.debug_clock(),
                  "what": "lnasr_debug_clock directly behind gmm_logprob (the dominant kernel) / forward on the same batch / idle"}
        b_tmp.free()
    except Exception as e:                                                # noqa: BLE001 -- a diagnostic, never the headline
        clocks = {"error": "%s: %s" % (type(e).__name__, e)}
    e2e = bw = None
    if not args.no_extras:
        note("end-to-end (host in, host out)")
        e2e, ok = end_to_end(rk, ctx, obs, resident_paths, step_s * 1e3, max(5, args.steps // 2))
        if not ok:
            sys.stderr.write("[rank %d] end-to-end paths differ from the resident run\n" % rank)
            rc = 1
        note("Baum-Welch statistics pass%s" % (" + all-reduce" if use_comm else ""))
        bw, ok = baum_welch(rk, ctx, use_comm)
        if not ok:
            sys.stderr.write("[rank %d] Baum-Welch statistics failed their checks\n" % rank)
            rc = 1
    refapi = None
    if not args.no_extras and world == 1 and not use_comm and T == 1000:
        note("reference API latency (one utterance through the mirror classes)")
        try:
            spec = importlib.util.spec_from_file_location("lnasr_refapi", os.path.join(REPO, "tools", "refapi.py"))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            refapi = mod.measure(ctx, obs=obs[0], check_path=want_head[0])
        except Exception as e:                                            # noqa: BLE001
            # an extra section: its failure is reported IN the line, it never takes the headline with it
            sys.stderr.write("[rank %d] reference API section failed: %s: %s\n" % (rank, type(e).__name__, e))
            refapi = {"error": "%s: %s" % (type(e).__name__, e)}
    full3 = None
    if not args.no_extras and world == 1 and not use_comm and U == UTT_PER_GPU and T == 1000:
        note("configs[3] in full on this GPU (8192 utterances)")
        try:
            full3, ok = full_config3(ctx, mdl, resident_paths)
            if not ok:
                sys.stderr.write("[rank %d] the 8192-utterance decode failed its checks\n" % rank)
                rc = 1
        except Exception as e:                                            # noqa: BLE001 -- an extra section, like reference_api
            sys.stderr.write("[rank %d] 8192-utterance section failed: %s: %s\n" % (rank, type(e).__name__, e))
            full3 = {"error": "%s: %s" % (type(e).__name__, e)}
    rc = int(max(rk.gather(rc)))
    note("done, rc=%d" % rc)

    if rank == 0:
        frames_total = n_ranks_rccl * U * T * args.steps
        value = frames_total / elapsed
        dom = max(prof.items(), key=lambda kv: kv[1][1])
        name, (launches, ms) = dom
        avg_s = ms / launches * 1e-3
        frames_launch = U * T * args.steps / launches
        if name.startswith("emit"):
            flop = FLOP_EMIT
        elif name.startswith("viterbi"):
            flop = FLOP_TRELLIS
        else:
            flop = FLOP_DECODE
        achieved = flop * frames_launch / avg_s / 1e12
        extra = {}
        if name == "emit_diag_i8":
            # the bank evaluates the quadratic form as a 56-bit fixed-point product on the int8 matrix cores (csrc/gmm_i8.hip):
            # 28 digit-plane products, 38 v_mfma_i32_16x16x64_i8 per 16 x 16 tile.  `achieved` / `frac` stay the ALGORITHMIC fp64 figure
            # against the fp64 matrix peak (the yardstick of every earlier round); the integer unit's own utilisation and
            # the digit-plane pass that precedes the bank (slice_features) are reported beside it.
            i8_ops = 38 * 64 * 2 * N_STATES * N_MIX                       # integer multiply-adds x 2 per frame
            sl = prof.get("slice_features")
            both = avg_s + (sl[1] / sl[0] * 1e-3 if sl else 0.0)
            rho, bank = model_range
            extra = {"matrix_unit": "int8 (v_mfma_i32_16x16x64_i8): 56-bit fixed point with one binary scale per GEMM row, per component "
                                    "and per frame; error ~3 * 2^-53 * rho on a component log-density at the frames it describes, "
                                    "models with rho > 2048 take the fp64 matrix cores instead (lnasr_model_range)",
                     "model_range_rho": rho, "integer_bank": bank,
                     "int8_ops_per_frame": i8_ops, "int8_achieved_tops": i8_ops * frames_launch / avg_s / 1e12,
                     "int8_peak_tops": PEAK_I8_TOPS, "int8_frac": i8_ops * frames_launch / avg_s / 1e12 / PEAK_I8_TOPS,
                     "with_slice_features": {"ms": both * 1e3, "achieved": flop * frames_launch / both / 1e12,
                                             "frac": flop * frames_launch / both / 1e12 / PEAK_FP64_TFLOPS}}
        kernels = {k: {"launches": n, "avg_ms": m / n} for k, (n, m) in sorted(prof.items(), key=lambda kv: -kv[1][1])}
        out = {
            "metric": METRIC,
            "value": value, "unit": "frames/s", "n_gpus": n_ranks_rccl, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "GMM-HMM Viterbi decode (emission + trellis + backtrace): N=128 states, M=16 diag "
                                   "mixtures, D=39, T=1000, %d utterances per GPU (BASELINE configs[3]: 8192 "
                                   "utterances over 8 GPUs)" % U,
                       "utterances_per_gpu": U, "frames_per_utterance": T, "sharding": "utterances, dp%d" % n_ranks_rccl,
                       "gather": gather, "parity": parity,
                       "inputs": "obs resident in HBM when the timed region starts (PCIe-inclusive: end_to_end)"},
            "roofline": {"bound": "mfma", "kernel": name, "achieved": achieved, "peak": PEAK_FP64_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_FP64_TFLOPS,
                         "traffic": _pmc_traffic(name)[0], "traffic_source": _pmc_traffic(name)[1],
                         "algorithmic_flop_per_frame": flop, "frames_per_launch": frames_launch,
                         "avg_launch_ms": avg_s * 1e3, "shader_clock": clocks, **extra,
                         "whole_step": {"flop_per_frame": FLOP_DECODE,
                                        "achieved_tflops": FLOP_DECODE * n_ranks_rccl * U * T / step_s / 1e12,
                                        "frac_of_fp64_peak": FLOP_DECODE * U * T / step_s / 1e12 / PEAK_FP64_TFLOPS,
                                        "hbm_bytes_per_frame": BYTES_DECODE,
                                        "frac_of_hbm_peak": BYTES_DECODE * U * T / step_s / 1e9 / PEAK_HBM_GBS}},
            "kernels": kernels,
        }
        if e2e is not None:
            out["end_to_end"] = e2e
            out["baum_welch"] = bw
        if refapi is not None:
            out["reference_api"] = refapi
        if full3 is not None:
            out["configs3_one_gpu"] = full3
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["cpu_baseline_reference_shaped"] = cpu_ref
            out["gpu_over_cpu"] = value / cpu["value"]
            out["gpu_over_cpu_note"] = "value / cpu_baseline.value: against the %d cores timed" % cpu["cores"]
            if cpu.get("single_socket_estimate"):
                out["gpu_over_single_socket_estimate"] = value / cpu["single_socket_estimate"]
        if rc == 0:
            print(json.dumps(out))
        sys.stdout.flush()
    rk.barrier()
    ctx.close()
    rk.close()
    return rc


if __name__ == "__main__":
    sys.exit(main())
