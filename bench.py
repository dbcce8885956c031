#!/usr/bin/env python3
"""Headline benchmark: frames/s of GMM-HMM Viterbi decode, N=128 states, M=16 mixtures, D=39
(BASELINE.json metric; workload = configs[3]: T=1000, 8192 utterances over 8 GPUs, i.e. 1024
utterances per GPU, weak scaling -- utterances shard with no data-path collective except the
final RCCL all-gather of the decoded state paths).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path (fp64-MFMA emission -> Viterbi trellis -> backtrace, then for
N > 1 the gather of paths) over this rank's 1024 utterances, with the observations already
resident in HBM.  Rank 0 prints ONE JSON line.  torch is used for rendezvous / barrier /
max-over-ranks only (gloo); the compute path is liblnasr_hip.so through ctypes.
"""
import os

for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")          # CPU-baseline workers: one BLAS thread per process

import argparse
import importlib.util
import json
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
PEAK_FP64_TFLOPS = 78.6                             # MI355X fp64 matrix = vector peak (AMD)
PEAK_HBM_GBS = 8000.0


def _load_oracle():
    spec = importlib.util.spec_from_file_location("lnasr_oracle", os.path.join(REPO, "oracle", "lnasr_oracle.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


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


def cpu_baseline():
    """Line B of BASELINE.md: best vectorised numpy form, one process per host core, one BLAS
    thread each, disjoint utterances; plus line A (reference-shaped loops, 1 core) as context."""
    import multiprocessing as mp
    avail = len(os.sched_getaffinity(0))
    # a 1-GPU box owns 16 of the host's logical CPUs (pool guidance); LNASR_CPU_WORKERS overrides
    cores = max(1, min(avail, int(os.environ.get("LNASR_CPU_WORKERS", "16"))))
    per = 40
    ctx = mp.get_context("fork")
    with ctx.Pool(cores) as pool:
        pool.map(_cpu_worker, [(5000 + i, 1) for i in range(cores)])          # warm caches / imports
        t0 = time.perf_counter()
        res = pool.map(_cpu_worker, [(6000 + i, per) for i in range(cores)])
        wall = time.perf_counter() - t0
    frames = cores * per * T_FRAMES
    out = {"value": frames / wall, "unit": "frames/s", "cores": cores, "kind": "port",
           "sample": "%d utterances x %d frames (same synthetic model/shape), oracle decode_diag_fused "
                     "(numpy GEMM emission + vectorised Viterbi), %d worker processes x 1 BLAS thread "
                     "(%d logical CPUs visible), %.1f s wall" % (cores * per, T_FRAMES, cores, avail, wall)}
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
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/pmc_traffic.json, corrected as MI355X_MICROARCH.md section HBM prescribes)."""
    p = os.path.join(REPO, "profiles", "pmc_traffic.json")
    try:
        with open(p) as fp:
            return json.load(fp).get(kernel, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--utts", type=int, default=UTT_PER_GPU, help="utterances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        args.gpus = world
    U, T = args.utts, T_FRAMES

    # CPU baseline first: fork-based workers must exist before this process touches the GPU
    cpu, cpu_ref = (None, None)
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu, cpu_ref = cpu_baseline()

    dist = None
    torch = None
    if world > 1:
        import torch
        import torch.distributed as dist
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    else:
        try:
            import torch
        except Exception:
            torch = None

    from lnasr import _ffi, synth
    ctx = _ffi.Context(local_rank)
    if torch is not None and torch.cuda.is_available():
        torch.cuda.set_device(local_rank % torch.cuda.device_count())

    def barrier():
        if dist is not None:
            dist.barrier()

    def sync():
        ctx.sync()
        if torch is not None and torch.cuda.is_available():
            torch.cuda.synchronize()

    mdl = synth.synth_model(N_STATES, N_MIX, N_DIM, seed=1)
    ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
    obs = synth.synth_obs(U, T, N_DIM, seed=1000 + rank)            # this rank's shard
    t0 = time.perf_counter()
    d_obs = ctx.to_device(obs)
    h2d_s = time.perf_counter() - t0
    d_path = ctx.empty((U, T), np.uint32)
    d_score = ctx.empty((U,), np.float64)
    d_all = None
    hung = False

    # RCCL communicator for the gather of paths (N > 1)
    gather = "n/a (single GPU)"
    if world > 1:
        import threading
        ok = 1
        state = {}

        def _init_comm():
            # communicator + one rehearsal gather; run under a watchdog so a wedged bootstrap cannot hang the bench
            try:
                ctx.comm_init(state["id"], rank, world)
                buf = ctx.empty((world, U, T), np.uint32)
                ctx.allgather_paths(d_path, N_STATES, out=buf)
                ctx.sync()
                state["buf"] = buf
            except Exception as e:                                   # noqa: BLE001
                state["err"] = e

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
            d_all = state["buf"]
        except Exception as e:                                       # noqa: BLE001
            ok = 0
            hung = True
            sys.stderr.write("[rank %d] RCCL init failed: %s\n" % (rank, e))
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            gather = "rccl all-gather of the state paths (u8 on the wire, u32 in memory), %d B per GPU per step" % (U * T)
        else:
            gather = "FAILED to initialise RCCL; timed without the gather"
            d_all = None

    def step():
        ctx.decode(d_obs, path_out=d_path, score_out=d_score)
        if d_all is not None:
            ctx.allgather_paths(d_path, N_STATES, out=d_all)

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

    # sanity: the decoded paths are valid state indices and the run is deterministic
    p = d_path.to_host()
    assert p.max() < N_STATES
    if d_all is not None:
        allp = d_all.to_host()
        assert np.array_equal(allp[rank], p), "gathered paths do not match the local shard"

    if rank == 0:
        frames_total = world * U * T * args.steps
        value = frames_total / elapsed
        dom = max(prof.items(), key=lambda kv: kv[1][1])
        name, (launches, ms) = dom
        avg_s = ms / launches * 1e-3
        frames_launch = U * T
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
            "metric": "frames/sec Viterbi decode, 128-state 16-mix GMM-HMM, D=39 MFCC",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "GMM-HMM Viterbi decode (emission + trellis + backtrace): N=128 states, M=16 diag "
                                   "mixtures, D=39, T=1000, %d utterances per GPU (BASELINE configs[3]: 8192 "
                                   "utterances over 8 GPUs)" % U,
                       "utterances_per_gpu": U, "frames_per_utterance": T, "sharding": "utterances, dp%d" % world,
                       "gather": gather, "inputs": "obs resident in HBM; H2D of this shard took %.1f ms "
                                                   "(%.2f GB/s) outside the timed region" % (
                                                       h2d_s * 1e3, obs.nbytes / h2d_s / 1e9)},
            "roofline": {"bound": "mfma", "kernel": name, "achieved": achieved, "peak": PEAK_FP64_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_FP64_TFLOPS,
                         "traffic": _pmc_traffic(name),
                         "algorithmic_flop_per_frame": flop, "frames_per_launch": frames_launch,
                         "avg_launch_ms": avg_s * 1e3,
                         "whole_step": {"flop_per_frame": FLOP_DECODE,
                                        "achieved_tflops": FLOP_DECODE * world * U * T / step_s / 1e12,
                                        "frac_of_fp64_peak": FLOP_DECODE * U * T / step_s / 1e12 / PEAK_FP64_TFLOPS,
                                        "hbm_bytes_per_frame": BYTES_DECODE,
                                        "frac_of_hbm_peak": BYTES_DECODE * U * T / step_s / 1e9 / PEAK_HBM_GBS}},
            "kernels": kernels,
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["cpu_baseline_reference_shaped"] = cpu_ref
            out["gpu_over_cpu"] = value / cpu["value"]
        print(json.dumps(out))
    barrier()
    if hung:
        sys.stdout.flush()
        os._exit(0)          # a wedged RCCL thread must not keep the process alive
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
