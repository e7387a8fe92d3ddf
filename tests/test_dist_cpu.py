"""N > 1 path on CPU: the landmark partition and the one exchange step (sum of the reduced camera
normal equations), exercised with torch.distributed / gloo at world_size 2 and 8. The arithmetic on each
rank is the CPU oracle (this is a test); the product's sharded path runs in tests/test_dist_gpu.py."""
import os
import socket

import numpy as np
import pytest

from svi_mapper_amd import dist as sdist
from svi_mapper_amd import synth


def test_landmark_shards_are_contiguous_and_balanced():
    rng = np.random.default_rng(0)
    n_lm = 1000
    deg = rng.integers(1, 30, n_lm)
    slots = np.repeat(np.arange(n_lm), deg)
    for n in (1, 2, 3, 8):
        b = sdist.landmark_shards(slots, n_lm, n)
        assert b[0] == 0 and b[-1] == n_lm and (np.diff(b) >= 0).all() and len(b) == n + 1
        per = [deg[b[r]:b[r + 1]].sum() for r in range(n)]
        assert max(per) - min(per) <= 2 * deg.max()
    # empty graph and more ranks than landmarks
    assert list(sdist.landmark_shards(np.zeros(0, np.int64), 0, 4)) == [0, 0, 0, 0, 0]
    b = sdist.landmark_shards(np.array([0, 0, 1]), 2, 4)
    assert b[0] == 0 and b[-1] == 2


def _reduced(orc, prob, lm_lo, lm_hi, with_pose_edges, lam):
    """local S, g of the landmarks [lm_lo, lm_hi) (and the pose-only edges if with_pose_edges)."""
    cam = prob["cam"]
    o = orc.OracleBA(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"])
    n_kf = prob["n_kf"]
    ID = synth.POSE_ID_SHIFT
    o.add_pose(ID, synth.pose12(prob["R_init"][0], prob["t_init"][0]), fixed=True)
    full = orc.OracleBA(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"])
    synth.build_ba_graph(full, prob)
    ty, pid, lid, z, info = full.get_edges()
    aty, ia, ib, az, ainfo = full.get_aux()
    for k in range(1, n_kf):
        o.add_pose(ID + k, synth.pose12(prob["R_init"][k], prob["t_init"][k]))
    if with_pose_edges:
        for t, a, b, zz, ii in zip(aty, ia, ib, az, ainfo):
            if t == 0:
                o.add_edge_se3(a, b, zz, ii)
            elif t == 1:
                o.add_edge_accel(a, zz[:3], None, ii[:6])
    keep = (lid >= lm_lo) & (lid < lm_hi)
    o.add_landmarks(np.arange(lm_lo, lm_hi), prob["lm_init"][lm_lo:lm_hi])
    o.add_edges_bulk(ty[keep], pid[keep], lid[keep], z[keep], info[keep], 1)
    o.initialize()
    H, b, pc, lc = o.dense_system()
    n = len(b)
    H = H + 0.0 * np.eye(n)
    nl = 3 * int((lc >= 0).sum())
    Hll = H[:nl, :nl] + lam * np.eye(nl)
    Hpl, Hpp = H[nl:, :nl], H[nl:, nl:]
    S = Hpp - Hpl @ np.linalg.solve(Hll, Hpl.T)
    g = b[nl:] - Hpl @ np.linalg.solve(Hll, b[:nl])
    return S, g, lid


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prob = synth.make_ba_problem(8, 120, 800, seed=9)
    lam = 2.5
    full = orc.OracleBA(*[prob["cam"][k] for k in ("fx", "fy", "cx", "cy", "baseline_m")])
    synth.build_ba_graph(full, prob)
    lid = full.get_edges()[2]
    bounds = sdist.landmark_shards(lid, prob["n_lm"], world)
    S, g, _ = _reduced(orc, prob, int(bounds[rank]), int(bounds[rank + 1]), rank == 0, lam)
    buf = torch.from_numpy(np.concatenate([S.reshape(-1), g]))
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)       # the path's one exchange step
    if rank == 0:
        Sref, gref, _ = _reduced(orc, prob, 0, prob["n_lm"], True, lam)
        n = len(gref)
        Ssum = buf[: n * n].numpy().reshape(n, n) + lam * np.eye(n)   # lambda on the pose diagonal once, after the sum
        q.put((float(np.abs(Ssum - (Sref + lam * np.eye(n))).max() / np.abs(Sref).max()),
               float(np.abs(buf[n * n:].numpy() - gref).max() / np.abs(gref).max())))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_gloo_sum_equals_unsharded(world):
    """world 8 = the split of the target machine (one shard per GPU of a node); several of the eight shards of this small graph
    hold a handful of landmarks only"""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    eS, eg = q.get(timeout=10)
    assert eS < 1e-12 and eg < 1e-12


def test_devptr_array_interface():
    d = sdist._DevPtr(0x1000, 17)
    ai = d.__cuda_array_interface__
    assert ai["shape"] == (17,) and ai["typestr"] == "<f8" and ai["data"] == (0x1000, False)


def test_dlpack_alias_states_its_device():
    """the capsule the hooks hand to torch aliases the caller's memory (no copy), keeps its bookkeeping until torch lets go,
    and never asks the runtime whose pointer it is (host memory here: the device case runs in the -m gpu tests)"""
    import gc
    import numpy as np
    import torch
    from svi_mapper_amd import _dlpack
    a = np.arange(12, dtype=np.float64)
    t = _dlpack.alias(a.ctypes.data, (3, 4), torch.float64, "cpu")
    assert t.shape == (3, 4) and t.dtype == torch.float64 and len(_dlpack._live) == 1
    t[1, 1] = -5.0
    assert a[5] == -5.0
    del t
    gc.collect()
    assert len(_dlpack._live) == 0
    assert _dlpack.alias(0, (0, 2), torch.float32, "cpu").shape == (0, 2)
