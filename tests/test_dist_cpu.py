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
