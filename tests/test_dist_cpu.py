"""N > 1 path on CPU: world_size-2 gloo processes exercise the sharding + the two collectives
(gather of paths, all-reduce of statistics) with the oracle standing in for the GPU compute."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import REPO, PKG_ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, PKG_ROOT)
    import importlib.util
    import torch.distributed as dist
    spec = importlib.util.spec_from_file_location("lnasr_oracle", os.path.join(REPO, "oracle", "lnasr_oracle.py"))
    oracle = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(oracle)
    from lnasr import synth
    from lnasr import dist as ldist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        U, T, N, M, D = 5, 30, 6, 3, 4                      # 5 utterances over 2 ranks: ragged shards (3 + 2)
        mdl = synth.synth_model(N, M, D, seed=2)
        obs = synth.synth_obs(U, T, D, seed=42)
        lo, hi = ldist.shard_range(U, rank, world)
        comm = ldist.GlooComm(rank, world)
        local = np.stack([oracle.decode_diag(obs[u], mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"])
                          for u in range(lo, hi)])
        allp = ldist.gather_paths(comm, local, U, T)
        st = oracle.bw_stats_diag([obs[u] for u in range(lo, hi)], mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"])
        packed = np.concatenate([np.ravel(st[k]) for k in ("xi_sum", "gamma_sum", "gamma0", "mix", "mix_x", "mix_xx")]
                                + [[st["loglik"], st["n_utt"]]])
        packed = ldist.allreduce_stats(comm, packed)
        q.put((rank, allp, packed))
    finally:
        dist.destroy_process_group()


def test_shard_range():
    sys.path.insert(0, PKG_ROOT)
    from lnasr import dist as ldist
    for n in (0, 1, 5, 8, 1024, 8192):
        for w in (1, 2, 3, 8):
            r = [ldist.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = ldist.shard_sizes(n, w)
            assert max(sizes) - min(sizes) <= 1
    assert ldist.shard_sizes(8192, 8) == [1024] * 8


def test_world2_gloo(oracle, synth):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    U, T, N, M, D = 5, 30, 6, 3, 4
    mdl = synth.synth_model(N, M, D, seed=2)
    obs = synth.synth_obs(U, T, D, seed=42)
    full = np.stack([oracle.decode_diag(obs[u], mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"]) for u in range(U)])
    st = oracle.bw_stats_diag(list(obs), mdl["A"], mdl["pi"], mdl["w"], mdl["mu"], mdl["var"])
    packed = np.concatenate([np.ravel(st[k]) for k in ("xi_sum", "gamma_sum", "gamma0", "mix", "mix_x", "mix_xx")]
                            + [[st["loglik"], st["n_utt"]]])
    for rank, allp, red in res:
        assert allp.dtype == np.uint32 and np.array_equal(allp, full)
        assert np.allclose(red, packed, rtol=1e-12, atol=1e-13)


def _run_bench(*argv, timeout=300):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + list(argv), capture_output=True, text=True,
                       timeout=timeout, env=env)
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    return r, lines


def test_bench_self_launch_two_ranks():
    """`bench.py --gpus 2` without a launcher starts two fresh ranks itself (torch.distributed.run), they
    rendezvous on 127.0.0.1 and rank 0 prints one line; --dry-launch keeps the rehearsal on the CPU."""
    r, lines = _run_bench("--gpus", "2", "--dry-launch")
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1
    out = lines[0]
    assert out["dry_launch"] and out["n_ranks"] == 2 and out["ranks"] == [0, 1] and out["local_ranks"] == [0, 1]
    assert out["distinct_pids"] == 2 and out["gather_ok"]
    assert "torch.distributed.run" in r.stderr and "--nproc-per-node 2" in r.stderr


def test_rank_started_by_a_foreign_launcher_gets_the_env_defaults():
    """A rank that somebody else's launcher started (WORLD_SIZE already set: bench.py's own launcher is skipped) must
    still see the environment this pool needs -- dmabuf IPC for RCCL, loopback interfaces, 127.0.0.1 rendezvous:
    bench.py sets them at import."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("HSA_ENABLE_IPC_MODE_LEGACY", "GLOO_SOCKET_IFNAME", "NCCL_SOCKET_IFNAME", "MASTER_ADDR")}
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--dry-launch"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "torch.distributed.run" not in r.stderr            # it did not start ranks of its own
    out = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")][0]
    seen = out["env"][0]
    assert seen["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert seen["GLOO_SOCKET_IFNAME"] == "lo" and seen["NCCL_SOCKET_IFNAME"] == "lo" and seen["MASTER_ADDR"] == "127.0.0.1"
    # and a value the launcher did set is left alone
    env["NCCL_SOCKET_IFNAME"] = "eth9"
    env["MASTER_PORT"] = str(_free_port())
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--dry-launch"],
                       capture_output=True, text=True, timeout=300, env=env)
    out = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")][0]
    assert out["env"][0]["NCCL_SOCKET_IFNAME"] == "eth9"


def test_bench_n1_path_never_imports_torch():
    """north_star: no PyTorch on the product path.  With one rank bench.py must not import torch at all (with N > 1
    only torch.distributed over gloo, for rendezvous).  Checked on the source: every torch import sits inside the
    Ranks / dry_launch helpers that a one-rank run without LNASR_BENCH_FORCE_COMM never enters."""
    import ast
    tree = ast.parse(open(os.path.join(REPO, "bench.py")).read())
    allowed = {"dry_launch", "__init__", "max"}
    for fn in ast.walk(tree):
        if isinstance(fn, (ast.FunctionDef, ast.AsyncFunctionDef)):
            for node in ast.walk(fn):
                if isinstance(node, (ast.Import, ast.ImportFrom)):
                    names = [a.name for a in node.names] + [getattr(node, "module", "") or ""]
                    if any(n.split(".")[0] == "torch" for n in names):
                        assert fn.name in allowed, fn.name
    for node in tree.body:                                      # nothing at module level
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            names = [a.name for a in node.names] + [getattr(node, "module", "") or ""]
            assert not any(n.split(".")[0] == "torch" for n in names)


def test_bench_launch_fails_loudly_without_gpus():
    """No GPU here: the two ranks must fail and the launcher must hand a non-zero exit code on -- no line with
    a number in it, no silent single-rank run."""
    from conftest import has_gpu
    if has_gpu():
        pytest.skip("a GPU is present")
    r, lines = _run_bench("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert not [l for l in lines if l.get("value")]
