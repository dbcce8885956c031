"""GPU, N > 1: the launch path of bench.py and the two RCCL collectives with more than one rank.

The two-rank parity test needs a lease with at least two GPUs and skips otherwise; the failure-path test runs
on any GPU box: with both ranks pinned to ONE device RCCL refuses the communicator, and the bench must then
time nothing, say so and exit non-zero.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO, PKG_ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra)
    return env


def test_bench_failed_communicator_times_nothing():
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--utts", "8", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600,
                       env=_clean_env(LNASR_BENCH_DEVICE="0", LNASR_COMM_TIMEOUT="90",
                                      PYTHONPATH=os.path.join(REPO, "tests", "_no_torch") + os.pathsep + os.environ.get("PYTHONPATH", "")))
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode != 0, r.stdout[-2000:]
    assert len(lines) == 1, (r.stdout[-2000:], r.stderr[-2000:])
    out = lines[0]
    assert out["value"] is None and out["ms_per_step"] is None and out["n_gpus"] == 0
    assert out["config"]["gather"].startswith("FAILED, nothing timed")


def test_bench_comm_path_one_rank_rehearsal():
    """The N > 1 code path of bench.py with one rank (LNASR_BENCH_FORCE_COMM): rendezvous group, unique id, RCCL
    communicator under the watchdog, RCCL gather inside the timed step, n_gpus from ncclCommCount -- everything but the
    links.  `torch` is made unimportable for the process: the path is framework-free."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "1", "--utts", "64",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600,
                       env=_clean_env(LNASR_BENCH_FORCE_COMM="1", LNASR_COMM_TIMEOUT="120",
                                      PYTHONPATH=os.path.join(REPO, "tests", "_no_torch") + os.pathsep + os.environ.get("PYTHONPATH", "")))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = lines[0]
    assert out["n_gpus"] == 1 and out["value"] > 0
    assert out["config"]["gather"].startswith("rccl all-gather of the state paths over 1 ranks")
    assert "pack_paths_u8" in out["kernels"] and "unpack_paths_u8" in out["kernels"]
    # the oracle check runs on the N > 1 path too, before anything is timed
    assert "bit-identical to the oracle's before timing" in out["config"]["parity"]
    # second collective of the path: the config-5 statistics pass + the RCCL all-reduce of the packed vector
    bw = out["baum_welch"]
    assert bw["bw_pass_ms"] > 0 and bw["bw_pass_f32_ms"] > 0 and bw["allreduce_ms"] > 0
    assert bw["allreduce_bytes"] == 8 * (256 * 256 + 2 * 256 + 256 * 32 * (1 + 2 * 39) + 3)
    assert bw["checks"].endswith(": ok")
    # host arrays in, host paths out, through the upload pipeline
    e2e = out["end_to_end"]
    assert e2e["value"] > 0 and e2e["h2d_GBps"] > 0 and e2e["paths_identical_to_resident_run"]
    assert e2e["pageable"]["paths_identical_to_resident_run"]


def test_bench_wrong_paths_time_nothing(tmp_path):
    """A kernel whose paths differ from the oracle's is not measured: with the checker's answer falsified (a test hook
    that flips one state of the expected paths) the bench prints the value-null line and exits non-zero."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "1", "--warmup", "0", "--utts", "8",
                        "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, timeout=600,
                       env=_clean_env(LNASR_BENCH_TEST_CORRUPT_ORACLE="1"))
    assert r.returncode != 0
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["value"] is None
    assert "differ from the oracle" in lines[0]["config"]["parity"]


def _rank(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, PKG_ROOT)
    from lnasr import _ffi, synth, rendezvous
    from lnasr import dist as ldist
    grp = rendezvous.TcpGroup(rank, world, timeout=240)         # the package's own rendezvous: no framework in a rank
    try:
        ctx = _ffi.Context(rank)
        comm = ldist.RcclComm(ctx, rank, world, grp.bcast(_ffi.comm_unique_id() if rank == 0 else None))
        out = {"info": ctx.comm_info()}
        N, M, D = 12, 4, 6
        mdl = synth.synth_model(N, M, D, seed=2)
        ctx.upload_gmmhmm(mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"], diag=True)
        for tag, U, T in (("bytes", 5, 32), ("words", 5, 31)):      # 3 + 2 utterances; 3*31 is odd: plain u32 route
            obs = synth.synth_obs(U, T, D, seed=42)
            lo, hi = ldist.shard_range(U, rank, world)
            local = ctx.decode(obs[lo:hi])
            sizes = ldist.shard_sizes(U, world)
            pad = np.zeros((max(sizes), T), dtype=np.uint32)
            pad[:hi - lo] = local
            allp = np.asarray(ctx.allgather_paths(pad, N)).reshape(world, max(sizes), T)
            out[tag] = np.concatenate([allp[r, :sizes[r]] for r in range(world)], axis=0)
            out[tag + "_generic"] = ldist.gather_paths(comm, local, U, T)
        obs = synth.synth_obs(5, 32, D, seed=42)
        lo, hi = ldist.shard_range(5, rank, world)
        st = ctx.bw_accumulate(obs[lo:hi])
        out["stats"] = ldist.allreduce_stats(comm, st.copy())
        out["mstep"] = ctx.bw_mstep(out["stats"])
        from lnasr.gmmhmm import GMMHMM
        gh = GMMHMM(N, M, D, mdl["A"].copy(), mdl["pi"].copy(), mdl["w"].copy(), mdl["mu"].copy(), mdl["var"].copy(), ctx=ctx)
        out["hist"] = gh.train_batch(obs[lo:hi], iters=2, eps=0.0, verbose=False, comm=comm)
        out["A_after"] = np.asarray(gh.A)
        q.put((rank, out))
        grp.barrier()
        ctx.close()
    finally:
        grp.close()


@pytest.mark.parametrize("world", [1, 2])
def test_rccl_gather_allreduce_and_training_loop(world, synth, oracle):
    """One process per GPU: sharded decode + RCCL gather (byte and u32 routes), statistics all-reduce, M-step and the
    whole training loop against the oracle / the single-process run.  world = 2 needs two GPUs and skips on a one-GPU
    box; world = 1 runs the same code over a one-rank communicator."""
    from lnasr import _ffi
    if _ffi.device_count() < world:
        pytest.skip("needs a lease with %d GPUs (one rank per GPU)" % world)
    import multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_rank, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    N, M, D = 12, 4, 6
    mdl = synth.synth_model(N, M, D, seed=2)
    for tag, U, T in (("bytes", 5, 32), ("words", 5, 31)):
        obs = synth.synth_obs(U, T, D, seed=42)
        want = np.stack([oracle.decode_diag(obs[u], mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"]) for u in range(U)])
        for r in range(world):
            assert res[r]["info"] == (r, world)
            assert np.array_equal(res[r][tag], want) and np.array_equal(res[r][tag + "_generic"], want)
    obs = synth.synth_obs(5, 32, D, seed=42)
    ref = oracle.bw_stats_diag(list(obs), mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"])
    packed = np.concatenate([np.ravel(ref[k]) for k in ("xi_sum", "gamma_sum", "gamma0", "mix", "mix_x", "mix_xx")])
    mo = oracle.mstep_from_stats(ref)
    for r in range(world):
        assert np.allclose(res[r]["stats"][:packed.size], packed, rtol=1e-9, atol=1e-12)
        assert np.array_equal(res[r]["stats"], res[0]["stats"])          # every rank holds the same sums
        for k in ("A", "pi", "w", "mu", "var"):
            assert np.allclose(res[r]["mstep"][k], mo[k], rtol=1e-8, atol=1e-10), k
    # the whole training loop over two ranks = the same loop over the whole batch on one
    from lnasr.gmmhmm import GMMHMM
    gh = GMMHMM(N, M, D, mdl["A"].copy(), mdl["pi"].copy(), mdl["w"].copy(), mdl["mu"].copy(), mdl["var"].copy())
    want = gh.train_batch(obs, iters=2, eps=0.0, verbose=False)
    for r in range(world):
        assert np.allclose(res[r]["hist"], want, rtol=1e-10) and np.allclose(res[r]["A_after"], gh.A, rtol=1e-8, atol=1e-10)
