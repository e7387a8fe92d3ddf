"""the product reads a .g2o file it did not write - tests/golden/handwritten_vi.g2o, produced without the
product's or the oracle's code, every tag of the reference's graphs in it incl. EDGE_SE3_LINEAR_ACCELERATION with a != 0
and a non-identity PARAMS_SE3OFFSET 3 - and must agree with the oracle, whose graph is built from the literals of the
same file by the tests' own reader (tests/g2o_text.py) through the oracle's API."""
import os

import numpy as np
import pytest

import g2o_text

pytestmark = pytest.mark.gpu

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "handwritten_vi.g2o")


def _pair(svi, oracle):
    lit = g2o_text.read(FIXTURE)
    fx, fy, cx, cy = lit["cams"][1]
    g = svi.BundleAdjuster(1.0, 1.0, 0.0, 0.0, 49.6325 / fx)     # the camera comes from PARAMS_CAMERACALIB 1
    g.load_g2o(FIXTURE)
    o = g2o_text.build(oracle.OracleBA(fx, fy, cx, cy, 49.6325 / fx), lit)
    return g, o, lit


def test_loaded_graph_is_the_file(svi, oracle):
    g, o, lit = _pair(svi, oracle)
    assert g.num_poses == o.num_poses == 4 and g.num_landmarks == o.num_landmarks == 13
    assert o.num_edges == g.num_edges == len(lit["proj"])       # (projection edges)
    st = g.stats()
    assert (st.n_edges_se3, st.n_edges_accel, st.n_edges_lmlm) == (3, 4, 1)
    ids_g, T_g = g.get_poses()
    ids_o, T_o = o.get_poses()
    assert np.array_equal(ids_g, ids_o)
    np.testing.assert_allclose(T_g, T_o, rtol=0, atol=1e-15)
    np.testing.assert_allclose(g.get_landmarks()[1], o.get_landmarks()[1], rtol=0, atol=0)


def test_errors_jacobians_chi2(svi, oracle):
    g, o, _ = _pair(svi, oracle)
    g.initialize()
    o.initialize()
    o.set_accel_numeric(False)
    for a, b in zip(g.edge_jacobians(), o.edge_jacobians()):
        np.testing.assert_allclose(a, b, rtol=1e-11, atol=1e-10)
    for a, b in zip(g.aux_jacobians(), o.aux_jacobians()):
        np.testing.assert_allclose(a, b, rtol=1e-11, atol=1e-12)
    cg, co = g.chi2(), o.chi2()
    assert abs(cg[0] - co[0]) <= 1e-11 * co[0] and abs(cg[1] - co[1]) <= 1e-11 * co[1]


@pytest.mark.parametrize("numeric", [True, False])
def test_lm_blocks(svi, oracle, numeric):
    """optimize(1) then a block of 10 (the reference's schedule): iteration counts, damping, estimates - against g2o's
    numeric gravity Jacobian within north_star's 1e-4, against the analytic one to rounding"""
    g, o, _ = _pair(svi, oracle)
    o.set_accel_numeric(numeric)
    g.initialize()
    o.initialize()
    tol = 1e-4 if numeric else 1e-8
    chi0 = g.chi2()[1]
    for n in (1, 10):
        assert g.optimize(n) == o.optimize(n)
        assert abs(g.lm_lambda - o.lm_lambda) <= tol * o.lm_lambda
        Tg, To = g.get_poses()[1], o.get_poses()[1]
        assert np.abs(Tg - To).max() <= tol * max(1.0, np.abs(To).max())
        pg, po = g.get_landmarks()[1], o.get_landmarks()[1]
        assert np.abs(pg - po).max() <= tol * np.abs(po).max()
        cg, co = g.chi2(), o.chi2()
        assert abs(cg[1] - co[1]) <= tol * co[1]
    assert g.chi2()[1] < 0.6 * chi0


def test_save_then_load_by_the_independent_reader(svi, oracle, tmp_path):
    """the writer against the tests' own reader: what svi_ba_save_g2o writes parses back to the same literals"""
    g, _, lit = _pair(svi, oracle)
    path = str(tmp_path / "out.g2o")
    g.save_g2o(path)
    back = g2o_text.read(path)
    assert back["fixed"] and set(back["fixed"]) == set(lit["fixed"])
    np.testing.assert_allclose(back["offsets"][3], lit["offsets"][3], atol=1e-15)
    np.testing.assert_allclose(back["cams"][1], lit["cams"][1], atol=0)
    assert sorted(p for p, _ in back["poses"]) == sorted(p for p, _ in lit["poses"])
    for (p1, id1, a1, i1), (p2, id2, a2, i2) in zip(back["accel"], lit["accel"]):
        assert (p1, id1) == (p2, id2) and np.array_equal(a1, a2) and np.array_equal(i1, i2)
    for e1, e2 in zip(back["proj"], lit["proj"]):
        assert e1[:4] == e2[:4] and np.array_equal(e1[4], e2[4]) and np.array_equal(e1[5], e2[5])
    for e1, e2 in zip(back["se3"], lit["se3"]):
        assert e1[:2] == e2[:2] and np.abs(e1[2] - e2[2]).max() < 1e-15 and np.array_equal(e1[3], e2[3])
