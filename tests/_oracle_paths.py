"""Helper of test_metric_config_full_size: the CPU oracle's Viterbi paths of the whole metric shard (N=128, M=16, D=39,
T=1000), decoded by a fork pool in a process of its own (no GPU is touched here).

    python tests/_oracle_paths.py OUT.npy N_UTTERANCES OBS_SEED WORKERS

Uses oracle.decode_diag_fused, which tests/test_oracle.py::test_metric_slice_fused_baseline pins to the reference's own
paths on the golden metric slice."""
import importlib.util
import multiprocessing as mp
import os
import sys

for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")

import numpy as np

TESTS = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(TESTS)
N, M, D, T = 128, 16, 39, 1000
_S = {}


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _work(args):
    lo, hi = args
    A, pi, Wc = _S["m"]
    return lo, np.stack([_S["o"].decode_diag_fused(_S["obs"][u], A, pi, Wc, N, M) for u in range(lo, hi)])


def main():
    out, U, seed, workers = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    # model and observations once, in the parent: the forked workers share the pages
    _S["o"] = _load("lnasr_oracle", os.path.join(REPO, "oracle", "lnasr_oracle.py"))
    synth = _load("lnasr_synth", os.path.join(REPO, "ln-asr_amd", "lnasr", "synth.py"))
    mdl = synth.synth_model(N, M, D, seed=1)
    _S["m"] = (mdl["A"], mdl["pi"], _S["o"].prep_diag_fused(mdl["w"], mdl["mu"], mdl["var"]))
    _S["obs"] = synth.synth_obs(U, T, D, seed=seed)
    step = 8
    jobs = [(lo, min(U, lo + step)) for lo in range(0, U, step)]
    paths = np.empty((U, T), dtype=np.uint32)
    with mp.get_context("fork").Pool(workers) as pool:
        for lo, p in pool.imap_unordered(_work, jobs):
            paths[lo:lo + len(p)] = p
    np.save(out, paths)


if __name__ == "__main__":
    main()
