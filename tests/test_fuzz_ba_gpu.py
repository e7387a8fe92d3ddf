"""Wider random sweep of the BA parity check: 40 seeds in the collected suite (7 s on an MI355X); more with
`python tests/test_fuzz_ba_gpu.py FIRST_SEED N`.

Every seed draws a graph shape (keyframes, landmarks, observations per landmark), a tile size and an
elimination order, runs the same LM blocks on the HIP path and on the CPU oracle and compares the
trajectories with the north-star tolerance (1e-4 relative)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import pytest  # noqa: E402

from svi_mapper_amd import synth  # noqa: E402

pytestmark = pytest.mark.gpu
REL = 1e-4


def _rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


def run_seed(svi, oracle, seed):
    r = np.random.default_rng(seed)
    n_kf = int(r.integers(3, 160))
    n_lm = int(r.integers(50, 6000))
    prob = synth.make_ba_problem(n_kf, n_lm, int(n_lm * r.uniform(2.5, 9)), seed=seed)
    cam = prob["cam"]
    kw = dict(chol_tile=int(r.choice([48, 96])), chol_order=int(r.integers(0, 2)))
    g = svi.BundleAdjuster(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"], **kw)
    o = oracle.OracleBA(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"])
    sg = synth.build_ba_graph(g, prob)
    so = synth.build_ba_graph(o, prob)
    np.testing.assert_array_equal(sg, so)
    g.initialize()
    o.initialize()
    blocks = (1, int(r.integers(2, 9)))
    for n in blocks:
        assert g.optimize(n) == o.optimize(n), (seed, "iterations performed differ")
    Tg, To = g.get_poses()[1], o.get_poses()[1]
    pg, po = g.get_landmarks()[1], o.get_landmarks()[1]
    errs = (_rel(Tg[:, 9:], To[:, 9:]), float(np.abs(Tg[:, :9] - To[:, :9]).max()), _rel(pg, po),
            abs(g.last_plain_chi2 - o.last_plain_chi2) / max(o.last_plain_chi2, 1e-300))
    g.close()
    return (n_kf, n_lm, [int(x) for x in np.ravel(sg)], kw, blocks), errs


@pytest.mark.parametrize("seed", range(1000, 1040))
def test_fuzz(svi, oracle, seed):
    shape, errs = run_seed(svi, oracle, seed)
    assert max(errs[:3]) < REL and errs[3] < 1e-6, (shape, errs)


if __name__ == "__main__":
    import svi_mapper_amd as svi_mod
    from oracle import oracle as oracle_mod
    oracle_mod.load()
    first, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000, 40)
    worst = 0.0
    for s in range(first, first + n):
        shape, errs = run_seed(svi_mod, oracle_mod, s)
        worst = max(worst, max(errs[:3]))
        print(s, shape, "t %.2e R %.2e lm %.2e chi2 %.2e" % errs, flush=True)
    print("worst", worst)
