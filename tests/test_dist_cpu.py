"""N > 1 path on CPU: world_size-2 processes exercise the sharding + the two collectives (gather of paths, all-reduce of
statistics) with the oracle standing in for the GPU compute -- once over gloo (the transport lives here, tests/_gloo_comm.py),
once over the package's own framework-free rendezvous (lnasr/rendezvous.py + lnasr.dist.TcpComm), which is also what
bench.py launches and meets its ranks with."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import REPO, PKG_ROOT, TESTS


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, transport="gloo"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, PKG_ROOT)
    sys.path.insert(0, TESTS)
    import importlib.util
    spec = importlib.util.spec_from_file_location("lnasr_oracle", os.path.join(REPO, "oracle", "lnasr_oracle.py"))
    oracle = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(oracle)
    from lnasr import synth
    from lnasr import dist as ldist

    class _Closer:
        def __init__(self, fn):
            self.destroy_process_group = fn

    if transport == "gloo":
        import torch.distributed as dist
        from _gloo_comm import GlooComm
        dist.init_process_group("gloo", rank=rank, world_size=world)
        comm = GlooComm(rank, world)
    else:
        from lnasr import rendezvous
        grp = rendezvous.TcpGroup(rank, world, timeout=120)
        comm = ldist.TcpComm(grp)
        dist = _Closer(grp.close)
    try:
        U, T, N, M, D = 5, 30, 6, 3, 4                      # 5 utterances over 2 ranks: ragged shards (3 + 2)
        mdl = synth.synth_model(N, M, D, seed=2)
        obs = synth.synth_obs(U, T, D, seed=42)
        lo, hi = ldist.shard_range(U, rank, world)
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


@pytest.mark.parametrize("transport", ["gloo", "tcp"])
def test_world2(transport, oracle, synth):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, transport)) for r in range(2)]
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


NO_TORCH = os.path.join(TESTS, "_no_torch")       # a `torch` package that raises on import: proves nobody imports it


def _poisoned(env):
    env = dict(env)
    env["PYTHONPATH"] = NO_TORCH + os.pathsep + env.get("PYTHONPATH", "")
    return env


def test_bench_self_launch_two_ranks():
    """`bench.py --gpus 2` without a launcher starts two fresh ranks of itself, they meet through the package's loopback
    rendezvous and rank 0 prints one line; --dry-launch keeps the rehearsal on the CPU.  Run with a `torch` that cannot be
    imported: no framework on the launch / rendezvous path."""
    import json
    import subprocess
    env = _poisoned({k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")})
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-launch"], capture_output=True,
                       text=True, timeout=300, env=env)
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1
    out = lines[0]
    assert out["dry_launch"] and out["n_ranks"] == 2 and out["ranks"] == [0, 1] and out["local_ranks"] == [0, 1]
    assert out["distinct_pids"] == 2 and out["gather_ok"]
    assert "started 2 ranks" in r.stderr


def test_ranks_of_a_foreign_launcher_meet_although_master_port_is_taken():
    """The driver starts the ranks with an elastic launcher whose own store LISTENS on MASTER_PORT.  The ranks must meet
    anyway (they never bind that port: rank 0 publishes an ephemeral one in the rendezvous file), with four ranks and a
    stale rendezvous file of an earlier run lying around."""
    import json
    import subprocess
    sys.path.insert(0, PKG_ROOT)
    from lnasr import rendezvous
    squatter = socket.socket()
    squatter.bind(("127.0.0.1", 0))
    squatter.listen(8)
    port = squatter.getsockname()[1]
    stale = rendezvous.rendezvous_file("127.0.0.1", str(port))
    with open(stale, "w") as fp:
        fp.write("1 deadbeefdeadbeef\n")                         # nobody listens on port 1
    base = _poisoned({k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")})
    procs = []
    try:
        for rk in (3, 1, 2, 0):                                   # rank 0 last: the others poll a stale file meanwhile
            env = dict(base, RANK=str(rk), LOCAL_RANK=str(rk), WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       LNASR_RDZV_TIMEOUT="120")
            procs.append((rk, subprocess.Popen([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "4", "--dry-launch"],
                                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)))
        outs = {rk: p.communicate(timeout=300) + (p.returncode,) for rk, p in procs}
    finally:
        squatter.close()
        for _, p in procs:
            if p.poll() is None:
                p.kill()
    assert all(o[2] == 0 for o in outs.values()), {rk: o[1][-500:] for rk, o in outs.items()}
    line = [json.loads(l) for l in outs[0][0].splitlines() if l.startswith("{")][0]
    assert line["n_ranks"] == 4 and line["ranks"] == [0, 1, 2, 3] and line["gather_ok"] and line["distinct_pids"] == 4
    assert not os.path.exists(stale)                              # rank 0 removes its file on the way out


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
    assert "started" not in r.stderr                          # it did not start ranks of its own
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


def test_no_framework_import_anywhere_in_bench_or_package():
    """north_star: no PyTorch.  Neither bench.py nor any module of the package imports torch (or any other framework) -- at
    module level or inside a function; checked on the source of every file."""
    import ast
    import glob
    files = [os.path.join(REPO, "bench.py"), os.path.join(REPO, "__graft_entry__.py")] + \
        sorted(glob.glob(os.path.join(PKG_ROOT, "**", "*.py"), recursive=True))
    assert len(files) > 10
    banned = {"torch", "jax", "tensorflow", "triton", "cupy"}
    for f in files:
        tree = ast.parse(open(f).read())
        for node in ast.walk(tree):
            if isinstance(node, (ast.Import, ast.ImportFrom)):
                names = [a.name for a in node.names] + [getattr(node, "module", "") or ""]
                assert not any(n.split(".")[0] in banned for n in names), (f, names)


def test_bench_launch_fails_loudly_without_gpus():
    """No GPU here: the two ranks must fail and the launcher must hand a non-zero exit code on -- no line with
    a number in it, no silent single-rank run."""
    from conftest import has_gpu
    if has_gpu():
        pytest.skip("a GPU is present")
    r, lines = _run_bench("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert not [l for l in lines if l.get("value")]


def test_rendezvous_wire_format_and_failure_modes(tmp_path):
    """The rendezvous of lnasr/rendezvous.py on its own: values survive the JSON wire format (bytes: the ncclUniqueId; numpy
    arrays; nesting), a rank without a rank 0 gives up with TimeoutError after its timeout instead of hanging, and a stranger
    talking nonsense to rank 0's port does not take a rank's place."""
    import threading
    import time
    sys.path.insert(0, PKG_ROOT)
    from lnasr import rendezvous as rz
    val = {"id": bytes(range(128)), "x": np.arange(6, dtype=np.uint32).reshape(2, 3), "f": [1.5, None, "s", np.float64(2.0)]}
    back = rz._dec(__import__("json").loads(__import__("json").dumps(rz._enc(val))))
    assert back["id"] == val["id"] and back["x"].dtype == np.uint32 and np.array_equal(back["x"], val["x"])
    assert back["f"] == [1.5, None, "s", 2.0]
    os.environ["LNASR_RDZV_FILE"] = str(tmp_path / "rdzv")
    try:
        t0 = time.time()
        with pytest.raises(TimeoutError):
            rz.TcpGroup(1, 2, timeout=1.0)                        # nobody is rank 0
        assert time.time() - t0 < 10
        out = {}

        def rank0():
            g = rz.TcpGroup(0, 2, timeout=60)
            out["gathered"] = g.allgather("zero")
            out["max"] = g.max(1.0)
            g.close()
        th = threading.Thread(target=rank0)
        th.start()
        for _ in range(200):                                      # wait for rank 0's file, then be rude to its port
            if os.path.exists(os.environ["LNASR_RDZV_FILE"]):
                break
            time.sleep(0.02)
        port = int(open(os.environ["LNASR_RDZV_FILE"]).read().split()[0])
        s = socket.create_connection(("127.0.0.1", port))
        s.sendall(b"\x05\x00\x00\x00hello")                       # a well-formed frame that is not JSON
        s.close()
        g1 = rz.TcpGroup(1, 2, timeout=60)
        assert g1.allgather("one") == ["zero", "one"] and g1.max(3.0) == 3.0
        g1.close()
        th.join(30)
        assert out == {"gathered": ["zero", "one"], "max": 3.0}
        assert not os.path.exists(os.environ["LNASR_RDZV_FILE"])
    finally:
        os.environ.pop("LNASR_RDZV_FILE", None)


def test_the_drivers_own_launcher_starts_ranks_that_meet():
    """The driver's command line: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...`.  That launcher's own store listens on P; the ranks it starts (which import no
    framework themselves) must rendezvous all the same.  CPU rehearsal with --dry-launch."""
    import json
    import subprocess
    pytest.importorskip("torch")                                   # the LAUNCHER is torch's; nothing of ours imports it
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "3",
                        "--dry-launch"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(out) == 1 and out[0]["n_ranks"] == 3 and out[0]["ranks"] == [0, 1, 2] and out[0]["gather_ok"]
    assert out[0]["distinct_pids"] == 3
