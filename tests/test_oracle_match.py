"""The matcher oracle against an independent numpy brute force and the committed golden vectors."""
import os

import numpy as np
import pytest

from svi_mapper_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _numpy_match(q, t, gate, cutoff):
    d = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(2).astype(np.int64)
    if gate is not None:
        quv, tuv = gate["q_uv"], gate["t_uv"]
        ok = (np.abs(tuv[None, :, 1] - quv[:, None, 1]) <= np.float32(gate["v_tol"])) & \
             (gate["q_umin"][:, None] <= tuv[None, :, 0]) & (tuv[None, :, 0] < gate["q_umax"][:, None])
        d = np.where(ok, d, 10**6)
    if d.shape[1] == 0:
        return np.full(len(q), -1, np.int32), np.full(len(q), 257, np.int32)
    idx = d.argmin(1)  # first minimum == lowest index
    dist = d[np.arange(len(q)), idx]
    miss = (dist >= cutoff) | (dist >= 10**6)
    return np.where(miss, -1, idx).astype(np.int32), np.where(miss, 257, dist).astype(np.int32)


@pytest.mark.parametrize("nq,nt,seed", [(64, 64, 1), (257, 1023, 2), (100, 3, 3)])
def test_oracle_vs_numpy(oracle, nq, nt, seed):
    s = synth.make_descriptor_pair(nq, nt, seed=seed) if nt > 8 else None
    if s is None:
        rng = np.random.default_rng(seed)
        q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
        t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
        i, d = oracle.match_hamming256(q, t)
        ri, rd = _numpy_match(q, t, None, 257)
    else:
        i, d = oracle.match_hamming256(s["q"], s["t"], s["gate"], s["cutoff"])
        ri, rd = _numpy_match(s["q"], s["t"], s["gate"], s["cutoff"])
    np.testing.assert_array_equal(i, ri)
    np.testing.assert_array_equal(d, rd)


def test_hamming_definition_lsb_popcount(oracle):
    """popcount(a ^ b) over 256 bits: the definition of src/types/CBNode.h:622-627."""
    a = np.zeros((3, 32), np.uint8)
    b = np.zeros((3, 32), np.uint8)
    b[0, 0] = 0x01
    b[1, 31] = 0x80
    b[2, :] = 0xFF
    np.testing.assert_array_equal(oracle.hamming256_pairs(a, b), [1, 1, 256])


def test_triangulation_formula_and_guard(oracle):
    cam = synth.kitti_camera()
    uvL = np.array([[700.0, 200.0], [700.0, 200.0], [700.0, 200.0]], np.float32)
    uvR = np.array([[690.0, 200.0], [699.995, 200.0], [700.0, 200.0]], np.float32)
    xyz, ok = oracle.triangulate_rectified(cam["fx"], cam["cx"], cam["cy"], cam["duR_flipped"], uvL, uvR)
    assert list(ok) == [1, 0, 0]  # 0.005 px and 0 px are below the 0.01 px guard (CTriangulator.h:21)
    z = cam["duR_flipped"] / 10.0
    np.testing.assert_allclose(xyz[0], [z * (700.0 - cam["cx"]) / cam["fx"], z * (200.0 - cam["cy"]) / cam["fx"], z], rtol=1e-14)


def test_golden_matcher_vectors(oracle):
    g = np.load(os.path.join(GOLD, "hamming_c2_small.npz"))
    gate = dict(q_uv=g["q_uv"], t_uv=g["t_uv"], q_umin=g["q_umin"], q_umax=g["q_umax"], v_tol=float(g["v_tol"]))
    i, d = oracle.match_hamming256(g["q"], g["t"], gate, int(g["cutoff"]))
    np.testing.assert_array_equal(i, g["idx"])
    np.testing.assert_array_equal(d, g["dist"])
    i, d = oracle.match_hamming256(g["q"], g["t"])
    np.testing.assert_array_equal(i, g["idx_ungated"])
    np.testing.assert_array_equal(d, g["dist_ungated"])
    t = np.load(os.path.join(GOLD, "triangulate.npz"))
    cam = synth.kitti_camera()
    xyz, ok = oracle.triangulate_rectified(cam["fx"], cam["cx"], cam["cy"], cam["duR_flipped"], t["uvL"], t["uvR"])
    np.testing.assert_array_equal(ok, t["ok"])
    np.testing.assert_array_equal(xyz, t["xyz"])
