#!/usr/bin/env python3
"""Headline benchmark: frames/s of GMM-HMM Viterbi decode, N=128 states, M=16 mixtures, D=39
(BASELINE.json metric; workload = configs[3]: T=1000, 8192 utterances over 8 GPUs, i.e. 1024
utterances per GPU, weak scaling -- utterances shard with no data-path collective except the
final RCCL all-gather of the decoded state paths).

    python bench.py [--gpus N] [--steps K] [--warmup W]

With --gpus N > 1 and no launcher environment (WORLD_SIZE unset) this process touches no GPU: it
starts N fresh ranks, one per GPU, as a child

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

forwards their output and returns the launcher's exit code.  Started by such a launcher (the
driver's own command line is the one above) the process is one rank.

A step = one pass of the hot path (fp64-MFMA emission -> Viterbi trellis -> backtrace, then for
N > 1 the RCCL gather of paths) over this rank's 1024 utterances, with the observations already
resident in HBM.  Rank 0 prints ONE JSON line.  torch is used for rendezvous / barrier /
max-over-ranks only (gloo); the compute path is liblnasr_hip.so through ctypes, the gather is
RCCL inside that library.  If the communicator or the rehearsal gather fails, NOTHING is timed:
rank 0 prints the line with "value": null and the failure in config.gather, and every rank exits
non-zero.

    --dry-launch    CPU rehearsal of the launch path (tests): the ranks rendezvous over gloo, shard
                    a toy batch, gather toy paths through the gloo transport and rank 0 prints a
                    {"dry_launch": true, "n_ranks": N, ...} line.  No GPU call, no timing.
"""
import os

for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")          # CPU-baseline workers: one BLAS thread per process

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
PEAK_HBM_GBS = 8000.0
METRIC = "frames/sec Viterbi decode, 128-state 16-mix GMM-HMM, D=39 MFCC"
N_CHECK = 4                                         # utterances compared with the oracle before timing


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
    """Start N fresh ranks (one per GPU) from a parent that has made no GPU call, wait, return their exit code."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this pool
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    sys.stderr.write("[bench] starting %d ranks: %s\n" % (args.gpus, " ".join(cmd)))
    sys.stderr.flush()
    return subprocess.call(cmd, env=env)


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


def cpu_baseline(check_obs):
    """Line B of BASELINE.md: best vectorised numpy form, one process per host core, one BLAS
    thread each, disjoint utterances; plus line A (reference-shaped loops, 1 core) as context.
    Also returns the oracle's paths of `check_obs` (the checker's answer for the pre-timing comparison)."""
    import multiprocessing as mp
    avail = len(os.sched_getaffinity(0))
    # a 1-GPU box owns 16 of the host's logical CPUs (pool guidance); LNASR_CPU_WORKERS overrides
    cores = max(1, min(avail, int(os.environ.get("LNASR_CPU_WORKERS", "16"))))
    per = int(os.environ.get("LNASR_CPU_UTTS", "400"))       # ~15 s per worker at ~27 k frames/s per core
    ctx = mp.get_context("fork")
    with ctx.Pool(cores) as pool:
        pool.map(_cpu_worker, [(5000 + i, 1) for i in range(cores)])          # warm caches / imports
        t0 = time.perf_counter()
        pool.map(_cpu_worker, [(6000 + i, per) for i in range(cores)])
        wall = time.perf_counter() - t0
    frames = cores * per * T_FRAMES
    cpu = _lscpu()
    import numpy
    out = {"value": frames / wall, "unit": "frames/s", "cores": cores, "kind": "port",
           "sample": "%d utterances x %d frames (same synthetic model/shape), oracle decode_diag_fused "
                     "(numpy %s GEMM emission + vectorised Viterbi), %d worker processes x 1 BLAS thread, %.1f s wall; "
                     "host: %s, %s socket(s) x %s cores x %s threads, %s logical CPUs (%d usable by this process)"
                     % (cores * per, T_FRAMES, numpy.__version__, cores, wall, cpu.get("model", "?"), cpu.get("sockets", "?"),
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
    Wc = o.prep_diag_fused(mdl["w"], mdl["mu"], mdl["var"])
    want = np.stack([o.decode_diag_fused(x, mdl["A"], mdl["pi"], Wc, N_STATES, N_MIX) for x in check_obs])
    return out, {"value": ref_shaped, "unit": "frames/s", "cores": 1,
                 "sample": "1 utterance x 200 frames, per-(state,mixture) density calls + per-(frame,state) "
                           "trellis updates (the reference's loop structure)"}, want


def _pmc_traffic(kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/pmc_traffic.json, corrected as MI355X_MICROARCH.md section HBM prescribes)."""
    p = os.path.join(REPO, "profiles", "pmc_traffic.json")
    try:
        with open(p) as fp:
            return json.load(fp).get(kernel, {}).get("hbm_bytes_per_launch")
    except Exception:                                                  # noqa: BLE001
        return None


# ---- --dry-launch: the launch path on CPU ----------------------------------------------------------------
def dry_launch(rank, world):
    import torch.distributed as dist
    from lnasr import dist as ldist
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        U, T = 4 * world + 1, 8                                           # ragged shards
        lo, hi = ldist.shard_range(U, rank, world)
        truth = (np.arange(U * T, dtype=np.uint32).reshape(U, T) * 7) % N_STATES
        comm = ldist.GlooComm(rank, world)
        allp = ldist.gather_paths(comm, truth[lo:hi], U, T)
        ok = bool(np.array_equal(allp, truth))
        seen = [None] * world
        dist.all_gather_object(seen, (rank, int(os.environ.get("LOCAL_RANK", "-1")), os.getpid()))
        if rank == 0:
            print(json.dumps({"dry_launch": True, "n_ranks": world, "ranks": sorted(s[0] for s in seen),
                              "local_ranks": sorted(s[1] for s in seen), "distinct_pids": len({s[2] for s in seen}),
                              "gather_ok": ok, "transport": "gloo (CPU rehearsal, nothing timed)"}))
            sys.stdout.flush()
        dist.barrier()
    finally:
        dist.destroy_process_group()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--utts", type=int, default=UTT_PER_GPU, help="utterances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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

    # CPU baseline first: fork-based workers must exist before this process touches the GPU
    cpu, cpu_ref, want_paths = (None, None, None)
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu, cpu_ref, want_paths = cpu_baseline(obs[:N_CHECK])

    dist = None
    torch = None
    # LNASR_BENCH_FORCE_COMM=1: take the N > 1 code path (gloo group, RCCL communicator under the watchdog, gather in
    # the timed step) with however many ranks there are -- on a one-GPU box a one-rank rehearsal of everything but
    # the links
    use_comm = world > 1 or bool(os.environ.get("LNASR_BENCH_FORCE_COMM"))
    if use_comm:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # one node: never depend on the hostname resolving
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")      # ... nor on which interface RCCL's bootstrap would pick
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    else:
        try:
            import torch
        except Exception:                                              # noqa: BLE001
            torch = None

    from lnasr import _ffi
    # one GPU per rank: LOCAL_RANK, and Context raises if that device does not exist.  LNASR_BENCH_DEVICE pins every
    # rank to one device -- only the failure-path test uses it (RCCL refuses two ranks on one GPU)
    dev = int(os.environ.get("LNASR_BENCH_DEVICE", local_rank))
    if "LNASR_BENCH_DEVICE" not in os.environ and world > 1 and _ffi.device_count() == 1:
        dev = 0             # a launcher that isolates one GPU per rank (HIP_VISIBLE_DEVICES): the rank's only device
    ctx = _ffi.Context(dev)
    if torch is not None and torch.cuda.is_available():
        torch.cuda.set_device(dev)

    def barrier():
        if dist is not None:
            dist.barrier()

    def sync():
        ctx.sync()
        if torch is not None and torch.cuda.is_available():
            torch.cuda.synchronize()

    mdl = synth.synth_model(N_STATES, N_MIX, N_DIM, seed=1)
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
    t0 = time.perf_counter()
    d_obs = ctx.to_device(obs)
    h2d_s = time.perf_counter() - t0
    d_path = ctx.empty((U, T), np.uint32)
    d_score = ctx.empty((U,), np.float64)
    d_all = None
    n_ranks_rccl = 1

    # RCCL communicator for the gather of paths (N > 1)
    gather = "n/a (single GPU)"
    if use_comm:
        import threading
        state = {}

        def _init_comm():
            # communicator + one rehearsal gather; under a watchdog so that a wedged bootstrap ends the run
            try:
                ctx.comm_init(state["id"], rank, world)
                buf = ctx.empty((world, U, T), np.uint32)
                ctx.allgather_paths(d_path, N_STATES, out=buf)
                ctx.sync()
                state["info"] = ctx.comm_info()
                state["buf"] = buf
            except Exception as e:                                   # noqa: BLE001
                state["err"] = e

        err = None
        try:
            ids = [_ffi.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            state["id"] = ids[0]
            th = threading.Thread(target=_init_comm, daemon=True)
            th.start()
            th.join(float(os.environ.get("LNASR_COMM_TIMEOUT", "180")))
            if th.is_alive():
                raise RuntimeError("RCCL bootstrap / first all-gather did not finish in time")
            if "err" in state:
                raise state["err"]
            if state["info"] != (rank, world):
                raise RuntimeError("RCCL reports rank/size %r, the launcher %r" % (state["info"], (rank, world)))
        except Exception as e:                                       # noqa: BLE001
            err = "%s: %s" % (type(e).__name__, e)
            sys.stderr.write("[rank %d] RCCL init failed: %s\n" % (rank, err))
        errs = [None] * world
        dist.all_gather_object(errs, err)
        if any(errs):
            # nothing is timed after a failed communicator (a wedged helper thread may still be inside the
            # library on this context): say so and leave non-zero, never through a re-exec
            if rank == 0:
                print(json.dumps({"metric": METRIC, "value": None, "unit": "frames/s", "n_gpus": 0, "steps": 0,
                                  "warmup": 0, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
                                  "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                                  "config": {"workload": "not run", "gather": "FAILED, nothing timed: %s" % (
                                      "; ".join("rank %d: %s" % (r, e) for r, e in enumerate(errs) if e))}}))
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(1)
        d_all = state["buf"]
        n_ranks_rccl = state["info"][1]
        gather = ("rccl all-gather of the state paths over %d ranks (ncclCommCount; u8 on the wire, u32 in memory), "
                  "%d B per GPU per step" % (n_ranks_rccl, U * T))

    def step():
        ctx.decode(d_obs, path_out=d_path, score_out=d_score)
        if d_all is not None:
            ctx.allgather_paths(d_path, N_STATES, out=d_all)

    # the checker's answer first: a fast kernel with different paths is not measured
    parity = "not checked (N > 1 or --no-cpu-baseline)"
    if want_paths is not None:
        step()
        sync()
        got = d_path.to_host()[:N_CHECK]
        if not np.array_equal(got, want_paths):
            sys.stderr.write("GPU paths differ from the oracle's on the bench's own input; nothing timed\n")
            return 1
        parity = "paths of the first %d utterances bit-identical to the oracle's before timing" % N_CHECK

    for _ in range(args.warmup):
        step()
    sync()
    barrier()
    ctx.prof_enable(True)
    ctx.prof_reset()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    barrier()
    ctx.prof_enable(False)
    prof = ctx.prof_report()

    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # sanity: the decoded paths are valid state indices and every rank holds every shard
    p = d_path.to_host()
    rc = 0
    if p.max() >= N_STATES:
        sys.stderr.write("[rank %d] decoded state index out of range\n" % rank)
        rc = 1
    if d_all is not None:
        allp = d_all.to_host()
        if not np.array_equal(allp[rank], p):
            sys.stderr.write("[rank %d] gathered paths do not match the local shard\n" % rank)
            rc = 1
        sums = [None] * world
        dist.all_gather_object(sums, int(p.astype(np.uint64).sum()))
        if [int(allp[r].astype(np.uint64).sum()) for r in range(world)] != sums:
            sys.stderr.write("[rank %d] gathered shards differ from what their owners decoded\n" % rank)
            rc = 1

    if rank == 0:
        frames_total = n_ranks_rccl * U * T * args.steps
        value = frames_total / elapsed
        dom = max(prof.items(), key=lambda kv: kv[1][1])
        name, (launches, ms) = dom
        avg_s = ms / launches * 1e-3
        frames_launch = U * T * args.steps / launches                    # a step may cut the batch into chunks
        if name.startswith("emit"):
            flop = FLOP_EMIT
        elif name.startswith("viterbi"):
            flop = FLOP_TRELLIS
        else:
            flop = FLOP_DECODE
        achieved = flop * frames_launch / avg_s / 1e12
        kernels = {k: {"launches": n, "avg_ms": m / n} for k, (n, m) in sorted(prof.items(), key=lambda kv: -kv[1][1])}
        step_s = elapsed / args.steps
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
                       "inputs": "obs resident in HBM; H2D of this shard took %.1f ms "
                                 "(%.2f GB/s) outside the timed region" % (h2d_s * 1e3, obs.nbytes / h2d_s / 1e9)},
            "roofline": {"bound": "mfma", "kernel": name, "achieved": achieved, "peak": PEAK_FP64_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_FP64_TFLOPS,
                         "traffic": _pmc_traffic(name),
                         "algorithmic_flop_per_frame": flop, "frames_per_launch": frames_launch,
                         "avg_launch_ms": avg_s * 1e3,
                         "whole_step": {"flop_per_frame": FLOP_DECODE,
                                        "achieved_tflops": FLOP_DECODE * n_ranks_rccl * U * T / step_s / 1e12,
                                        "frac_of_fp64_peak": FLOP_DECODE * U * T / step_s / 1e12 / PEAK_FP64_TFLOPS,
                                        "hbm_bytes_per_frame": BYTES_DECODE,
                                        "frac_of_hbm_peak": BYTES_DECODE * U * T / step_s / 1e9 / PEAK_HBM_GBS}},
            "kernels": kernels,
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["cpu_baseline_reference_shaped"] = cpu_ref
            out["gpu_over_cpu"] = value / cpu["value"]
        if rc == 0:
            print(json.dumps(out))
        sys.stdout.flush()
    barrier()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
