"""GPU parity of the bundle adjustment against the CPU oracle.

Tolerance (BASELINE.json north_star): poses/landmarks within 1e-4 relative after the same LM
iteration schedule. Per-edge quantities and the reduced system are compared much tighter."""
import numpy as np
import pytest

from svi_mapper_amd import synth

pytestmark = pytest.mark.gpu

REL = 1e-4


def _make(cls, prob, **kw):
    cam = prob["cam"]
    ba = cls(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"], **kw)
    stored = synth.build_ba_graph(ba, prob)
    return ba, stored


def _rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


@pytest.fixture(scope="module")
def small():
    return synth.make_ba_problem(12, 300, 2200, seed=7)


def test_graph_rules_match(svi, oracle, small):
    g, sg = _make(svi.BundleAdjuster, small)
    o, so = _make(oracle.OracleBA, small)
    np.testing.assert_array_equal(sg, so)
    assert sg.sum() > 1000 and (sg > 0).all()  # all three edge kinds occur


def test_edge_jacobians(svi, oracle, small):
    g, _ = _make(svi.BundleAdjuster, small)
    o, _ = _make(oracle.OracleBA, small)
    g.initialize()
    eg, Jpg, Jlg = g.edge_jacobians()
    eo, Jpo, Jlo = o.edge_jacobians()
    assert eg.shape == eo.shape
    np.testing.assert_allclose(eg, eo, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(Jpg, Jpo, rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(Jlg, Jlo, rtol=1e-11, atol=1e-9)


def test_initial_chi2(svi, oracle, small):
    g, _ = _make(svi.BundleAdjuster, small)
    o, _ = _make(oracle.OracleBA, small)
    g.initialize()
    o.initialize()
    pg, rg = g.chi2()
    po, ro = o.chi2()
    assert abs(pg - po) <= 1e-10 * po and abs(rg - ro) <= 1e-10 * ro


@pytest.mark.parametrize("tile", [48, 96])
def test_reduced_system_is_schur_of_oracle_H(svi, oracle, small, tile):
    g, _ = _make(svi.BundleAdjuster, small, chol_tile=tile)
    o, _ = _make(oracle.OracleBA, small)
    g.initialize()
    o.initialize()
    lam = 3.7
    S, gv = g.reduced_system(lam)
    H, b, pc, lc = o.dense_system()
    n = len(b)
    H = H + lam * np.eye(n)
    nl = 3 * (lc >= 0).sum()
    Hll, Hpl, Hpp = H[:nl, :nl], H[nl:, :nl], H[nl:, nl:]
    Sref = Hpp - Hpl @ np.linalg.solve(Hll, Hpl.T)
    gref = b[nl:] - Hpl @ np.linalg.solve(Hll, b[:nl])
    assert S.shape == Sref.shape
    scale = np.abs(Sref).max()
    assert np.abs(S - Sref).max() <= 1e-10 * scale
    assert np.abs(gv - gref).max() <= 1e-10 * np.abs(gref).max()


@pytest.mark.parametrize("tile", [48, 96])
def test_single_iteration_matches(svi, oracle, small, tile):
    g, _ = _make(svi.BundleAdjuster, small, chol_tile=tile)
    o, _ = _make(oracle.OracleBA, small)
    g.initialize()
    o.initialize()
    assert g.optimize(1) == 1 and o.optimize(1) == 1
    assert abs(g.lm_lambda - o.lm_lambda) <= 1e-9 * o.lm_lambda
    _, Tg = g.get_poses()
    _, To = o.get_poses()
    _, pg = g.get_landmarks()
    _, po = o.get_landmarks()
    assert _rel(Tg, To) < 1e-9 and _rel(pg, po) < 1e-9
    assert abs(g.last_plain_chi2 - o.last_plain_chi2) <= 1e-8 * o.last_plain_chi2


def test_more_poses_than_one_workgroup_has_lanes(svi, oracle):
    """700 key frames: the trial poses (an extra workgroup of the one-launch backward substitution, one lane per pose) and the
    XCD-wise placement of the Schur work both loop / split where config 4 (500 key frames) does not."""
    prob = synth.make_ba_problem(700, 5000, 30000, seed=0x2BC)
    g, _ = _make(svi.BundleAdjuster, prob)
    o, _ = _make(oracle.OracleBA, prob)
    g.initialize()
    o.initialize()
    for n in (1, 3):
        assert g.optimize(n) == o.optimize(n)
    _, Tg = g.get_poses()
    _, To = o.get_poses()
    _, pg = g.get_landmarks()
    _, po = o.get_landmarks()
    assert _rel(Tg, To) < 1e-9 and _rel(pg, po) < 1e-9
    assert abs(g.last_plain_chi2 - o.last_plain_chi2) <= 1e-8 * o.last_plain_chi2


def _full_schedule(svi, oracle, prob, **kw):
    g, _ = _make(svi.BundleAdjuster, prob, **kw)
    o, _ = _make(oracle.OracleBA, prob)
    g.initialize()
    o.initialize()
    ng, eg = g.optimize_until()
    no, eo = o.optimize_until()
    ids_g, Tg = g.get_poses()
    ids_o, To = o.get_poses()
    lid_g, pg = g.get_landmarks()
    lid_o, po = o.get_landmarks()
    np.testing.assert_array_equal(ids_g, ids_o)
    np.testing.assert_array_equal(lid_g, lid_o)
    return dict(g=g, o=o, iters=(ng, eg, no, eo), T=(Tg, To), p=(pg, po))


def test_full_schedule_small(svi, oracle, small):
    r = _full_schedule(svi, oracle, small)
    ng, eg, no, eo = r["iters"]
    assert (ng, eg) == (no, eo)  # same _optimizeUnLimited block structure and executed LM iterations
    Tg, To = r["T"]
    pg, po = r["p"]
    assert _rel(Tg[:, 9:], To[:, 9:]) < REL          # translations
    assert np.abs(Tg[:, :9] - To[:, :9]).max() < REL  # rotation matrix entries
    assert _rel(pg, po) < REL
    cg, co = r["g"].last_plain_chi2, r["o"].last_plain_chi2
    assert abs(cg - co) <= 1e-6 * co


def test_full_schedule_medium_tile96(svi, oracle):
    """(the default tile edge is 48 since round 2: the other full-schedule tests run with it, this one keeps 96 covered)"""
    prob = synth.make_ba_problem(40, 4000, 30000, seed=11)
    r = _full_schedule(svi, oracle, prob, chol_tile=96)
    assert r["iters"][:2] == r["iters"][2:]
    Tg, To = r["T"]
    pg, po = r["p"]
    assert _rel(Tg[:, 9:], To[:, 9:]) < REL and np.abs(Tg[:, :9] - To[:, :9]).max() < REL and _rel(pg, po) < REL


@pytest.mark.parametrize("tile", [48, 96])
def test_elimination_orders_agree(svi, tile):
    """nested-dissection order (independent chains, level-scheduled launches) against the natural single chain:
    same reduced system (the debug tap reports natural order), same LM trajectory to rounding"""
    prob = synth.make_ba_problem(120, 6000, 50000, seed=5)
    res = {}
    for order in (0, 1):
        g, _ = _make(svi.BundleAdjuster, prob, chol_tile=tile, chol_order=order)
        g.initialize()
        S, gv = g.reduced_system(1e-3)
        n_it = g.optimize(6)
        _, T = g.get_poses()
        _, p = g.get_landmarks()
        st = g.stats()
        res[order] = (S, gv, n_it, T, p, g.last_plain_chi2, int(st.chol_steps), int(st.chol_tiles_nnz), int(st.chol_n))
        g.close()
    a, b = res[0], res[1]
    assert np.abs(a[0] - b[0]).max() <= 1e-9 * np.abs(b[0]).max() and np.abs(a[1] - b[1]).max() <= 1e-9 * np.abs(b[1]).max()
    assert a[2] == b[2]
    assert np.abs(a[3] - b[3]).max() < 1e-8 and _rel(a[4], b[4]) < 1e-8 and abs(a[5] - b[5]) <= 1e-9 * b[5]
    nt = -(-a[8] // tile)
    assert b[6] == nt                       # natural order: one chain, one level per tile column
    assert a[6] < b[6], (a[6], b[6])        # nested dissection: fewer launches on the critical path


@pytest.mark.parametrize("seed", [21, 22, 23, 24, 25, 26])
def test_random_problems(svi, oracle, seed):
    """further random graphs of different shapes, both tile sizes and elimination orders: same LM trajectory as the oracle"""
    r = np.random.default_rng(seed)
    n_kf = int(r.integers(8, 70))
    n_lm = int(r.integers(200, 3000))
    prob = synth.make_ba_problem(n_kf, n_lm, int(n_lm * r.uniform(4, 8)), seed=seed)
    g, sg = _make(svi.BundleAdjuster, prob, chol_tile=int(r.choice([48, 96])), chol_order=int(r.integers(0, 2)))
    o, so = _make(oracle.OracleBA, prob)
    np.testing.assert_array_equal(sg, so)
    g.initialize()
    o.initialize()
    for n in (1, 7):
        assert g.optimize(n) == o.optimize(n)
    _, Tg = g.get_poses()
    _, To = o.get_poses()
    _, pg = g.get_landmarks()
    _, po = o.get_landmarks()
    assert _rel(Tg[:, 9:], To[:, 9:]) < REL and np.abs(Tg[:, :9] - To[:, :9]).max() < REL and _rel(pg, po) < REL
    assert abs(g.last_plain_chi2 - o.last_plain_chi2) <= 1e-6 * o.last_plain_chi2


def _nonlinear_graph(cls, tau, rot, seed=3, n_free=6, n_lm=80):
    """non-robust inverse-depth edges, badly rotated initial poses and (with a tiny tau) almost no damping: Gauss-Newton
    steps that overshoot"""
    cam = synth.kitti_camera()
    r = np.random.default_rng(seed)
    ba = cls(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"], lm_tau=tau)
    ident = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0.0])
    ba.add_pose(1000000, ident, True)

    def rotm(w):
        th = np.linalg.norm(w)
        k = w / th
        K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K

    t_true = [np.array([0.05 * (k + 1), 0, 0.8 * (k + 1)]) for k in range(n_free)]
    for k in range(n_free):
        ba.add_pose(1000001 + k, np.concatenate([rotm(r.normal(0, rot, 3)).ravel(), t_true[k] + r.normal(0, 1.0, 3)]), False)
    pts = np.stack([r.uniform(-4, 4, n_lm), r.uniform(-1.5, 1.5, n_lm), 0.8 * n_free + r.uniform(4, 25, n_lm)], 1)
    for l in range(n_lm):
        ba.add_landmark(l, pts[l] + r.normal(0, 0.5, 3))
        for k in range(n_free + 1):
            pc = pts[l] - (np.zeros(3) if k == 0 else t_true[k - 1])
            z = np.array([cam["fx"] * pc[0] / pc[2] + cam["cx"], cam["fy"] * pc[1] / pc[2] + cam["cy"], 1.0 / pc[2]])
            ba.add_edges_bulk([2], [1000000 + k], [l], z[None], np.array([[1.0, 0, 0, 1.0, 0, 1000.0]]), [0])
    return ba


def test_failed_factorisation_is_a_failed_trial(svi, oracle, small):
    """an odometry edge with NEGATIVE information makes the reduced camera system indefinite until lambda has grown: the
    tile Cholesky has to report the non-positive pivot (it is recognised after the sweep, from the pivots the sweep stored),
    the trial counts as failed like g2o's CHOLMOD 'not positive definite', and the LM sequence follows the oracle"""
    def build(cls):
        ba, _ = _make(cls, small)
        Z = np.concatenate([np.eye(3).ravel(), [0.0, 0.0, 0.8]])
        info = np.zeros(21)
        info[[0, 6, 11, 15, 18, 20]] = -3e7      # the diagonal of the upper triangle
        ba.add_edge_se3(1000005, 1000006, Z, info, robust=False)
        ba.initialize()
        return ba
    g, o = build(svi.BundleAdjuster), build(oracle.OracleBA)
    for n in (1, 2, 3):
        assert g.optimize(n) == o.optimize(n)
        st = g.stats()
        assert st.lm_trials == o.trials and st.lm_iterations == o.iterations
        assert abs(g.lm_lambda - o.lm_lambda) <= 1e-6 * o.lm_lambda
    assert st.chol_failures > 0, "the graph was supposed to provoke failed factorisations"
    assert _rel(g.get_poses()[1], o.get_poses()[1]) < REL


@pytest.mark.parametrize("rot,seed", [(0.3, 3), (0.3, 9), (0.25, 1), (0.35, 6), (0.35, 8)])
def test_rejected_trials_follow_g2o(svi, oracle, rot, seed):
    """trials are rejected (lambda *= nu, nu *= 2, state restored) before one is accepted: the accept / reject sequence,
    the damping and the iteration counts follow the oracle. The comparison stops while chi2 is still far above the rounding
    floor (there the sign of rho is noise on both sides), and the undamped Gauss-Newton steps amplify the last-bit
    differences of the two summation orders, hence 1e-4 on lambda instead of the usual 1e-6."""
    g = _nonlinear_graph(svi.BundleAdjuster, 1e-12, rot, seed=seed)
    o = _nonlinear_graph(oracle.OracleBA, 1e-12, rot, seed=seed)
    g.initialize()
    o.initialize()
    for n in (1, 3, 3):
        assert g.optimize(n) == o.optimize(n)
        st = g.stats()
        assert st.lm_trials == o.trials and st.lm_iterations == o.iterations
        assert abs(g.lm_lambda - o.lm_lambda) <= 1e-4 * o.lm_lambda
    assert st.lm_trials > st.lm_iterations, "the graph was supposed to provoke rejected trials"
    cg, co = g.chi2(), o.chi2()
    assert abs(cg[0] - co[0]) <= 1e-4 * co[0]
    pg, po = g.get_poses()[1], o.get_poses()[1]
    assert np.abs(pg - po).max() <= 1e-4 * np.abs(po).max()  # north_star's relative bound (translations are ~100 m)


def test_full_information_matrices(svi, oracle):
    """edges with non-diagonal information (a .g2o graph may carry them; the reference itself only sets diagonals): the
    six-plane kernels instead of the diagonal specialisation"""
    prob = synth.make_ba_problem(14, 400, 2600, seed=31)
    r = np.random.default_rng(9)
    res = []
    for cls in (svi.BundleAdjuster, oracle.OracleBA):
        ba, _ = _make(cls, prob)
        k = 150
        lm = r.permutation(prob["n_lm"])[:k] if cls is svi.BundleAdjuster else lm
        kf = (r.integers(1, prob["n_kf"], k)).astype(np.int64) if cls is svi.BundleAdjuster else kf
        # avoid duplicating an existing (pose, landmark) edge: use landmarks ids offset into fresh landmarks
        new_ids = 500000 + np.arange(k)
        if cls is svi.BundleAdjuster:
            P = prob["lm_true"][lm] + r.normal(0, 0.05, (k, 3))
            A = r.normal(0, 1, (k, 3, 3))
            O = np.einsum("nij,nkj->nik", A, A) + 5 * np.eye(3)
            info = np.stack([O[:, 0, 0], O[:, 0, 1], O[:, 0, 2], O[:, 1, 1], O[:, 1, 2], O[:, 2, 2]], 1)
            z = np.einsum("nji,nj->ni", prob["R_true"][kf], prob["lm_true"][lm] - prob["t_true"][kf]) + r.normal(0, 0.01, (k, 3))
            kf2 = np.maximum(kf - 1, 0)
            z2 = np.einsum("nji,nj->ni", prob["R_true"][kf2], prob["lm_true"][lm] - prob["t_true"][kf2]) + r.normal(0, 0.01, (k, 3))
        ba.add_landmarks(new_ids, P)
        ba.add_edges_bulk(np.zeros(k, np.int32), 1000000 + kf, new_ids, z, info, np.ones(k, np.int32))
        ba.add_edges_bulk(np.zeros(k, np.int32), 1000000 + kf2, new_ids, z2, info, np.zeros(k, np.int32))
        ba.initialize()
        n = ba.optimize(5)
        res.append((n, ba.get_poses()[1], ba.get_landmarks()[1], ba.chi2()))
    (ng, Tg, pg, cg), (no, To, po, co) = res
    assert ng == no
    assert _rel(Tg[:, 9:], To[:, 9:]) < REL and np.abs(Tg[:, :9] - To[:, :9]).max() < REL and _rel(pg, po) < REL
    assert abs(cg[0] - co[0]) <= 1e-6 * co[0] and abs(cg[1] - co[1]) <= 1e-6 * co[1]


def test_loop_closure_tracks_break_the_band(svi, oracle):
    """landmarks re-observed by key frames far away (a revisit): the reduced system is no longer banded, the order
    search has to cope (crossing tracks only add dependencies or fall back to the natural order); oracle parity"""
    prob = synth.make_ba_problem(90, 3000, 24000, seed=13)
    r = np.random.default_rng(2)
    last_seen = np.full(prob["n_lm"], -1)
    np.maximum.at(last_seen, prob["obs_lm"], prob["obs_kf"])
    early = np.nonzero((last_seen >= 0) & (last_seen < prob["n_kf"] - 30))[0]
    extra_lm = r.choice(early, 60, replace=False)
    extra_kf = (prob["n_kf"] - 1 - r.integers(0, 10, 60)).astype(np.int64)     # seen again from the last ten key frames
    res = []
    for cls, kw in ((svi.BundleAdjuster, dict(chol_tile=96)), (oracle.OracleBA, {})):
        ba, _ = _make(cls, prob, **kw)
        Rt, tt = prob["R_true"], prob["t_true"]
        z = np.einsum("nji,nj->ni", Rt[extra_kf], prob["lm_true"][extra_lm] - tt[extra_kf])
        info = np.tile(np.array([10.0, 0, 0, 10.0, 0, 10.0]), (60, 1))
        ba.add_edges_bulk(np.zeros(60, np.int32), 1000000 + extra_kf, extra_lm, z, info, np.ones(60, np.int32))
        ba.initialize()
        n = ba.optimize(5)
        res.append((n, ba.get_poses()[1], ba.get_landmarks()[1], ba.chi2()[0]))
        if cls is svi.BundleAdjuster:
            st = ba.stats()
            assert st.chol_steps <= -(-st.chol_n // 48)
    (ng, Tg, pg, cg), (no, To, po, co) = res
    assert ng == no
    assert _rel(Tg[:, 9:], To[:, 9:]) < REL and np.abs(Tg[:, :9] - To[:, :9]).max() < REL and _rel(pg, po) < REL
    assert abs(cg - co) <= 1e-6 * co


def test_c3_first_block_parity(svi, oracle):
    """BASELINE config 3 (100 KF / 20 k landmarks / 150 k edges): optimize(1) + optimize(10)."""
    prob = synth.make_c3()
    g, sg = _make(svi.BundleAdjuster, prob)
    o, so = _make(oracle.OracleBA, prob)
    np.testing.assert_array_equal(sg, so)
    assert abs(int(sg.sum()) - 150000) < 1500
    g.initialize()
    o.initialize()
    for n in (1, 10):
        assert g.optimize(n) == o.optimize(n)
    _, Tg = g.get_poses()
    _, To = o.get_poses()
    _, pg = g.get_landmarks()
    _, po = o.get_landmarks()
    assert _rel(Tg[:, 9:], To[:, 9:]) < REL and np.abs(Tg[:, :9] - To[:, :9]).max() < REL and _rel(pg, po) < REL
    assert abs(g.last_plain_chi2 - o.last_plain_chi2) <= 1e-6 * o.last_plain_chi2


def _assert_full_schedule_parity(r):
    """north_star: the same LM iteration count, poses / landmarks within 1e-4 relative, plain chi2 within 1e-6."""
    ng, eg, no, eo = r["iters"]
    assert (ng, eg) == (no, eo), "nominal / executed LM iterations: product %s, oracle %s" % ((ng, eg), (no, eo))
    Tg, To = r["T"]
    pg, po = r["p"]
    assert _rel(Tg[:, 9:], To[:, 9:]) < REL
    assert np.abs(Tg[:, :9] - To[:, :9]).max() < REL
    assert _rel(pg, po) < REL
    cg, co = r["g"].last_plain_chi2, r["o"].last_plain_chi2
    assert abs(cg - co) <= 1e-6 * co
    st = r["g"].stats()
    assert st.chol_failures == 0 and st.lm_iterations == eg


def test_c3_full_schedule_parity(svi, oracle):
    """BASELINE config 3, the WHOLE _optimizeUnLimited schedule (Cg2oOptimizer.cpp:954-980: optimize(1), then blocks of
    optimize(10) while the plain chi2 improves by more than 1 %): 51 LM iterations in product and oracle alike."""
    r = _full_schedule(svi, oracle, synth.make_c3())
    _assert_full_schedule_parity(r)
    assert r["iters"][1] >= 11  # more than the first block: the ratio test of :969 has been taken at least once


def test_c4_full_schedule_parity(svi, oracle):
    """BASELINE config 4 at full size (500 KF / 100 k landmarks / 800 k edges), the whole schedule: 61 LM iterations. The
    oracle's full-system sparse LL' needs about half a second per iteration on one host core, so this is the longest test
    of the suite (about a minute)."""
    r = _full_schedule(svi, oracle, synth.make_c4())
    _assert_full_schedule_parity(r)
    assert r["iters"][1] >= 21


def test_reduced_system_after_a_block_that_ended_on_a_rejected_trial(svi):
    """(round-2 advisor finding) With lm_max_trials = 1 a rejected first trial ends the block while the sweep that was
    speculated behind it - the linearisation of the REJECTED state - sits in the linearisation buffers.  The next
    linearisation outside optimize() (the reduced-system tap) must not mistake it for its own: after initialize() the tap has
    to return the system of the untouched initial estimate, bit for bit what a fresh handle returns."""
    import functools
    found = 0
    for rot, seed in [(0.3, 3), (0.35, 6), (0.35, 8), (0.4, 2), (0.4, 5), (0.45, 9)]:
        ref = _nonlinear_graph(svi.BundleAdjuster, 1e-12, rot, seed=seed)
        ref.initialize()
        S0, g0 = ref.reduced_system(0.25)
        g = _nonlinear_graph(functools.partial(svi.BundleAdjuster, lm_max_trials=1), 1e-12, rot, seed=seed)
        g.initialize()
        T0 = g.get_poses()[1].copy()
        done = g.optimize(3)
        st = g.stats()
        if not (done == 1 and st.lm_trials == 1 and np.array_equal(g.get_poses()[1], T0)):
            continue  # the first trial was accepted: not the case under test
        found += 1
        g.initialize()
        S1, g1 = g.reduced_system(0.25)
        assert np.array_equal(S0, S1) and np.array_equal(g0, g1)
        assert g.optimize(1) == 1  # and the handle goes on working
    assert found > 0, "none of the graphs had its first trial rejected"


def test_backsolve_timeout_is_an_internal_error(svi):
    """A hand-over of the one-launch backward substitution that never arrives is a defect, not a property of the matrix:
    svi_ba_optimize returns SVI_ERR_INTERNAL (9), the trial is not counted as a failed factorisation, lambda does not
    grow, and with the spin budget restored the same handle produces what a fresh handle does."""
    from svi_mapper_amd import _capi
    lib = _capi.load_library()
    prob = synth.make_ba_problem(100, 3000, 20000, seed=77)
    ref, _ = _make(svi.BundleAdjuster, prob)
    ref.initialize()
    assert ref.stats().chol_steps >= 3
    assert ref.optimize(3) == 3
    g, _ = _make(svi.BundleAdjuster, prob)
    g.initialize()
    try:
        assert lib.svi_debug_set_backsolve_spin_limit(1) == 0
        with pytest.raises(svi.SviError) as e:
            g.optimize(3)
        assert e.value.status == 9
    finally:
        assert lib.svi_debug_set_backsolve_spin_limit(1 << 22) == 0
    st = g.stats()
    assert st.chol_failures == 0 and st.backsolve_timeouts == 1 and st.lm_iterations == 0
    assert g.optimize(3) == 3
    assert np.array_equal(g.get_poses()[1], ref.get_poses()[1]) and g.lm_lambda == ref.lm_lambda


def test_long_trajectory_takes_the_per_level_backward_substitution(svi, oracle):
    """2100 key frames = 263 tile columns: more than the one-launch backward substitution may hold resident (one workgroup per
    column, bounded by the number of compute units), so the per-level launches run - against the oracle."""
    prob = synth.make_ba_problem(2100, 6000, 30000, seed=0x2100)
    g, _ = _make(svi.BundleAdjuster, prob)
    o, _ = _make(oracle.OracleBA, prob)
    g.initialize()
    o.initialize()
    assert g.stats().chol_n // 48 > 256
    for n in (1, 2):
        assert g.optimize(n) == o.optimize(n)
    Tg, To = g.get_poses()[1], o.get_poses()[1]
    pg, po = g.get_landmarks()[1], o.get_landmarks()[1]
    assert _rel(Tg[:, 9:], To[:, 9:]) < REL and np.abs(Tg[:, :9] - To[:, :9]).max() < REL and _rel(pg, po) < REL
    assert abs(g.last_plain_chi2 - o.last_plain_chi2) <= 1e-6 * o.last_plain_chi2


@pytest.mark.parametrize("overlap", [True, False])
def test_staged_reduction_equals_single_launch(svi, oracle, overlap, monkeypatch):
    """The staged Schur reduction (work lists cut by the dependency level of a tile's column, tiles summed
    inside k_schur by the wave that delivers a cell's last slab, the factorisation reading S + S_upd and - with `overlap` - running
    level by level on the main stream behind stream waits on memory values while k_schur is still reducing the later stages on a
    second stream) is off by default; switched on, it must reproduce the single-launch path: same LM iterations, same estimates to
    round-off, and the oracle's within the north_star tolerance."""
    prob = synth.make_ba_problem(160, 12000, 90000, seed=0x51A6)
    ref, _ = _make(svi.BundleAdjuster, prob)
    ref.initialize()
    assert ref.stats().chol_steps >= 5
    done_ref = [ref.optimize(n) for n in (1, 4)]
    Tr, pr = ref.get_poses()[1], ref.get_landmarks()[1]
    monkeypatch.setenv("SVI_SCHUR_STAGES", "1,2,4")
    if not overlap:
        monkeypatch.setenv("SVI_NO_OVERLAP", "1")
    g, _ = _make(svi.BundleAdjuster, prob)
    g.initialize()
    monkeypatch.delenv("SVI_SCHUR_STAGES")
    done = [g.optimize(n) for n in (1, 4)]
    assert done == done_ref and g.stats().chol_failures == 0
    Tg, pg = g.get_poses()[1], g.get_landmarks()[1]
    assert _rel(Tg, Tr) < 1e-9 and _rel(pg, pr) < 1e-9
    assert abs(g.last_plain_chi2 - ref.last_plain_chi2) <= 1e-9 * ref.last_plain_chi2
    o, _ = _make(oracle.OracleBA, prob)
    o.initialize()
    assert [o.optimize(n) for n in (1, 4)] == done
    assert _rel(Tg[:, 9:], o.get_poses()[1][:, 9:]) < REL and _rel(pg, o.get_landmarks()[1]) < REL


def test_fixed_and_closure_edges(svi, oracle, small):
    """Landmark-closure EdgePointXYZ with a fixed partner (Cg2oOptimizer.cpp:445-458) and a fixed landmark."""
    def build(cls):
        ba, _ = _make(cls, small)
        ba.add_landmark(900001, small["lm_init"][5] + np.array([0.05, -0.02, 0.1]), fixed=True)
        ba.add_edge_lm_lm(900001, 5, np.zeros(3), 1000 * np.array([1, 0, 0, 1, 0, 1.0]), robust=True)
        ba.add_landmark(900002, small["lm_init"][9] + 0.01, fixed=True)
        ba.add_edge_lm_lm(17, 900002, np.array([0.01, 0, 0]), 1000 * np.array([1, 0, 0, 1, 0, 1.0]), robust=True)
        ba.initialize()
        return ba
    g, o = build(svi.BundleAdjuster), build(oracle.OracleBA)
    for n in (1, 5):
        assert g.optimize(n) == o.optimize(n)
    _, pg = g.get_landmarks()
    _, po = o.get_landmarks()
    _, Tg = g.get_poses()
    _, To = o.get_poses()
    assert _rel(pg, po) < 1e-7 and _rel(Tg, To) < 1e-7
    np.testing.assert_array_equal(g.get_landmark(900001), o.get_landmark(900001))  # fixed stays put


def test_write_back_rules(svi, oracle, small):
    """_applyOptimizationToLandmarks / ToKeyFrames (Cg2oOptimizer.cpp:1468-1540): estimate - shift, diverged
    landmarks (|p|^2 >= 1e12) leave the graph with their edges; checked against the oracle's graph after the same rule"""
    g, _ = _make(svi.BundleAdjuster, small)
    o, _ = _make(oracle.OracleBA, small)
    for ba in (g, o):
        ba.add_landmark(777777, [2e6, 1.0, 0.0])           # diverged, no edges
        ba.initialize()
        ba.optimize(2)
    shift = np.array([10.0, -2.5, 0.125])
    ids_o, p_o = o.get_landmarks()
    idk_o, T_o = o.get_poses()
    nl, ne = g.num_landmarks, g.num_edges
    out = g.apply_optimization(shift)
    assert out["erased"] == 1 and o.prune_diverged() == 1
    assert g.num_landmarks == nl - 1 == o.num_landmarks and g.num_edges == ne == o.num_edges
    assert np.array_equal(out["lm_ids"], ids_o) and np.array_equal(out["kf_ids"], idk_o)
    gone = out["lm_ids"] == 777777
    assert np.array_equal(out["lm_kept"], (~gone).astype(np.uint8)) and np.all(out["lm_xyz"][gone] == 0)
    assert _rel(out["lm_xyz"][~gone], p_o[~gone] - shift) < REL
    assert np.abs(out["kf_T"][:, :9] - T_o[:, :9]).max() < REL and _rel(out["kf_T"][:, 9:], T_o[:, 9:] - shift) < REL
    # the write-back is an exact subtraction of what the getters report
    ids_g, p_g = g.get_landmarks()
    keep = ~gone
    assert np.array_equal(ids_g, out["lm_ids"][keep]) and np.array_equal(p_g - shift, out["lm_xyz"][keep])
    g.initialize()                                           # the pruned graph optimises on
    assert g.optimize(1) == 1


def _tiny_graph(cls, n_free, n_lm, seed, fix_all=False):
    """hand-built graph: pose 1000000 fixed at the origin, n_free poses along +z, landmarks seen by every pose"""
    r = np.random.default_rng(seed)
    cam = synth.kitti_camera()
    ba = cls(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"])
    ident = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0.0])
    ba.add_pose(1000000, ident, True)
    for k in range(n_free):
        T = ident.copy()
        T[9:] = [0.02 * (k + 1), 0.0, 0.8 * (k + 1)]
        ba.add_pose(1000001 + k, T + np.concatenate([np.zeros(9), r.normal(0, 0.02, 3)]), fix_all)
    pts = np.stack([r.uniform(-3, 3, n_lm), r.uniform(-1, 1, n_lm), 0.8 * n_free + r.uniform(6, 20, n_lm)], 1)
    for l in range(n_lm):
        ba.add_landmark(l, pts[l] + r.normal(0, 0.05, 3))
        for k in range(n_free + 1):
            pc = pts[l] - np.array([0.02 * k, 0.0, 0.8 * k])
            z = pc + r.normal(0, 0.01, 3)
            ba.add_edges_bulk([0], [1000000 + k], [l], z[None], (1000.0 / z[2] * np.array([1, 0, 0, 1, 0, 1.0]))[None], [1])
    return ba


@pytest.mark.parametrize("n_free,n_lm,fix_all", [(1, 3, False), (0, 5, False), (3, 1, False), (2, 6, True), (17, 40, False)])
def test_degenerate_and_tiny_graphs(svi, oracle, n_free, n_lm, fix_all):
    """one free pose, no free pose at all (pure landmark refinement), a single landmark, every pose fixed, and a system
    of exactly one tile plus one pose: same LM trajectory as the oracle"""
    g = _tiny_graph(svi.BundleAdjuster, n_free, n_lm, 3, fix_all)
    o = _tiny_graph(oracle.OracleBA, n_free, n_lm, 3, fix_all)
    g.initialize()
    o.initialize()
    assert g.optimize(4) == o.optimize(4)
    _, Tg = g.get_poses()
    _, To = o.get_poses()
    _, pg = g.get_landmarks()
    _, po = o.get_landmarks()
    assert np.abs(Tg - To).max() < 1e-7 and np.abs(pg - po).max() < 1e-7
    cg, co = g.chi2(), o.chi2()
    assert abs(cg[0] - co[0]) <= 1e-6 * max(1.0, co[0]) and abs(cg[1] - co[1]) <= 1e-6 * max(1.0, co[1])
    # run to convergence: at the noise floor the gain ratio's sign is rounding noise, so only the block structure
    # (nominal count) and the converged state are compared, not the number of accepted steps
    ng, no = g.optimize_until(), o.optimize_until()
    assert ng[0] == no[0]
    cg, co = g.chi2(), o.chi2()
    assert abs(cg[0] - co[0]) <= 1e-6 * max(1.0, co[0])
    assert np.abs(g.get_landmarks()[1] - o.get_landmarks()[1]).max() < 1e-6


def test_unsupported_shapes_fail_loudly(svi, small):
    ba, _ = _make(svi.BundleAdjuster, small)
    ba.add_edge_lm_lm(3, 4, np.zeros(3), np.array([1, 0, 0, 1, 0, 1.0]))
    with pytest.raises(svi.SviError) as ei:
        ba.initialize()
    assert ei.value.status == 5
    ba2, _ = _make(svi.BundleAdjuster, small)
    with pytest.raises(svi.SviError):
        ba2.optimize(1)  # before initialize


def test_g2o_round_trip(svi, small, tmp_path):
    a, _ = _make(svi.BundleAdjuster, small)
    f = tmp_path / "graph.g2o"
    a.save_g2o(f)
    cam = small["cam"]
    b = svi.BundleAdjuster(1, 1, 0, 0, cam["baseline_m"])
    b.load_g2o(f)
    a.initialize()
    b.initialize()
    assert a.optimize(3) == b.optimize(3)
    _, Ta = a.get_poses()
    _, Tb = b.get_poses()
    assert _rel(Ta, Tb) < 1e-9  # quaternion text round trip of the initial rotations


def test_c4_first_iterations_parity(svi, oracle):
    """BASELINE config 4 at FULL size against the oracle: optimize(1) + optimize(2) (the oracle's full-system sparse
    factorisation takes about half a second per iteration, so three iterations is what fits a test) - the nine-level
    tile Cholesky, the cell-based Schur reduction and the sharded work lists at the size the benchmark is quoted on."""
    prob = synth.make_c4()
    g, sg = _make(svi.BundleAdjuster, prob)
    o, so = _make(oracle.OracleBA, prob)
    np.testing.assert_array_equal(sg, so)
    g.initialize()
    o.initialize()
    for n in (1, 2):
        assert g.optimize(n) == o.optimize(n)
    _, Tg = g.get_poses()
    _, To = o.get_poses()
    _, pg = g.get_landmarks()
    _, po = o.get_landmarks()
    assert _rel(Tg[:, 9:], To[:, 9:]) < REL and np.abs(Tg[:, :9] - To[:, :9]).max() < REL and _rel(pg, po) < REL
    assert abs(g.last_plain_chi2 - o.last_plain_chi2) <= 1e-6 * o.last_plain_chi2
    assert g.stats().chol_steps >= 5  # the level-scheduled path, not a single chain


def test_c4_properties(svi):
    """BASELINE config 4 (500 KF / 100 k landmarks / 800 k edges), size-independent properties:
    accepted LM steps never increase the robust chi2 and the result is reproducible run to run (the 2-way landmark-sharded
    solve at this size is tests/test_dist_gpu.py::test_c4_two_shards_equal_unsharded)."""
    prob = synth.make_c4()
    g, sg = _make(svi.BundleAdjuster, prob)
    assert abs(int(sg.sum()) - 800000) < 8000
    g.initialize()
    _, r0 = g.chi2()
    chis = [r0]
    for _ in range(3):
        assert g.optimize(1) == 1
        chis.append(g.chi2()[1])
    assert all(b <= a * (1 + 1e-12) for a, b in zip(chis, chis[1:])) and chis[-1] < 0.9 * chis[0]
    st = g.stats()
    assert st.chol_failures == 0 and st.chol_n == 6 * 499
    _, T1 = g.get_poses()
    g2, _ = _make(svi.BundleAdjuster, prob)
    g2.initialize()
    for _ in range(3):
        g2.optimize(1)
    _, T2 = g2.get_poses()
    assert _rel(T1, T2) < 1e-9
