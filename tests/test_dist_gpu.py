"""The landmark-sharded BA on real kernels: two shards of one graph, their reduced systems summed
through the all-reduce hook, must reproduce the unsharded solve."""
import threading

import numpy as np
import pytest

from svi_mapper_amd import synth

pytestmark = pytest.mark.gpu


def _run_sharded(svi, prob, n_ranks, iters, extra=None, make=None, until=False):
    """n_ranks handles in n_ranks threads of one process; the hook sums their buffers in a fixed order.
    make(rank, n_ranks) (optional) builds the handle with its graph instead of synth.build_ba_graph(prob)."""
    import torch
    cam = prob["cam"] if prob is not None else None
    barrier = threading.Barrier(n_ranks)
    slots = [None] * n_ranks
    out = [None] * n_ranks
    errs = []

    def hook_for(rank):
        def hook(ptr, count, stream):
            from svi_mapper_amd import _dlpack
            ext = torch.cuda.ExternalStream(stream)
            t = _dlpack.alias(ptr, (count,), torch.float64, "cuda:0")
            ext.synchronize()
            slots[rank] = t
            barrier.wait()
            if rank == 0:
                total = slots[0].clone()
                for r in range(1, n_ranks):
                    total += slots[r]
                for r in range(n_ranks):
                    slots[r].copy_(total)
                torch.cuda.synchronize()
            barrier.wait()
            return 0
        return hook

    def work(rank):
        try:
            if make is not None:
                ba = make(rank, n_ranks)
            else:
                ba = svi.BundleAdjuster(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"], rank=rank, n_ranks=n_ranks)
                synth.build_ba_graph(ba, prob)
            if extra is not None:
                extra(ba)
            ba.set_allreduce(hook_for(rank))
            ba.initialize()
            done, lams = [], []
            if until:   # the whole _optimizeUnLimited schedule (every block's lambda_0 exchange included)
                done = list(ba.optimize_until())
                lams.append(ba.lm_lambda)
            for n in ([] if until else iters):
                done.append(ba.optimize(n))
                lams.append(ba.lm_lambda)
            st = ba.stats()
            out[rank] = (done, ba.get_poses()[1], ba.get_landmarks()[1], ba.chi2(), st.n_landmarks_local, st.lm_trials, st.chol_failures,
                         lams)
            ba.close()
        except Exception as e:  # noqa: BLE001
            errs.append(e)
            barrier.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(n_ranks)]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
    if errs:
        raise errs[0]
    return out


@pytest.mark.parametrize("n_ranks", [2, 3, 8])
def test_sharded_equals_unsharded(svi, n_ranks):
    prob = synth.make_ba_problem(30, 2000, 15000, seed=21)
    cam = prob["cam"]
    iters = (1, 5)
    ref = svi.BundleAdjuster(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"])
    synth.build_ba_graph(ref, prob)
    ref.initialize()
    done = [ref.optimize(n) for n in iters]
    _, T, = ref.get_poses()
    _, p = ref.get_landmarks()
    out = _run_sharded(svi, prob, n_ranks, iters)
    locals_ = [o[4] for o in out]
    assert sum(locals_) == prob["n_lm"] and min(locals_) > 0
    for o in out:
        assert o[0] == done
        assert np.abs(o[1] - T).max() < 1e-9
        assert np.abs(o[2] - p).max() < 1e-9 * max(1.0, np.abs(p).max())  # every rank ends up with ALL landmarks
        assert abs(o[3][0] - ref.chi2()[0]) <= 1e-9 * ref.chi2()[0]


def test_c4_two_shards_equal_unsharded(svi):
    """BASELINE config 4 at full size (500 key frames / 100 k landmarks / 800 k edges) cut into two landmark shards on one GPU,
    their reduced systems summed through the hook: the same iterations, the same estimates as the unsharded solve"""
    prob = synth.make_c4()
    cam = prob["cam"]
    iters = (1, 3)
    ref = svi.BundleAdjuster(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"])
    synth.build_ba_graph(ref, prob)
    ref.initialize()
    done = [ref.optimize(n) for n in iters]
    T, p = ref.get_poses()[1], ref.get_landmarks()[1]
    chi = ref.chi2()
    ref.close()
    out = _run_sharded(svi, prob, 2, iters)
    assert sum(o[4] for o in out) == prob["n_lm"] and abs(out[0][4] - out[1][4]) < 0.05 * prob["n_lm"]   # balanced by edge count
    for o in out:
        assert o[0] == done
        assert np.abs(o[1] - T).max() < 1e-9 and np.abs(o[2] - p).max() < 1e-9 * np.abs(p).max()
        assert abs(o[3][0] - chi[0]) <= 1e-9 * chi[0]
    assert np.array_equal(out[0][1], out[1][1])


def test_c4_eight_shards_equal_unsharded(svi):
    """BASELINE config 4 as the target machine cuts it: EIGHT landmark shards (here on one GPU, one thread per rank). Rank 0
    alone holds the pose-only edges, lambda_0 is exchanged at the first iteration of each block, the later iterations keep
    their pose sums local: the same LM iterations / trials on every rank, bit-identical poses between the ranks, and the
    estimates of the unsharded solve."""
    prob = synth.make_c4()
    cam = prob["cam"]
    iters = (1, 3)
    ref = svi.BundleAdjuster(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"])
    synth.build_ba_graph(ref, prob)
    ref.initialize()
    done = [ref.optimize(n) for n in iters]
    T, p = ref.get_poses()[1], ref.get_landmarks()[1]
    chi = ref.chi2()
    st = ref.stats()
    ref.close()
    out = _run_sharded(svi, prob, 8, iters)
    loc = [o[4] for o in out]
    assert sum(loc) == prob["n_lm"] and min(loc) > 0 and max(loc) < 1.6 * min(loc)   # (shards are balanced by EDGE count, not landmarks)
    for o in out:
        assert o[0] == done and o[5] == st.lm_trials and o[6] == 0
        assert o[7] == out[0][7]                                  # the same damping on every rank, bit for bit
        assert np.array_equal(o[1], out[0][1])
        assert np.abs(o[1] - T).max() < 1e-9 and np.abs(o[2] - p).max() < 1e-9 * np.abs(p).max()
        assert abs(o[3][0] - chi[0]) <= 1e-9 * chi[0]


def test_c4_eight_shards_full_schedule(svi):
    """BASELINE config 4 in eight landmark shards through the WHOLE _optimizeUnLimited schedule (61 LM iterations: seven
    lambda_0 exchanges, sixty iterations whose pose sums stay local): the same nominal / executed counts as the unsharded
    solve, the same estimates (the shards' partial sums are added in another order than the single handle's: 1e-8, not bits),
    and bit-identical poses and damping between the eight ranks."""
    prob = synth.make_c4()
    cam = prob["cam"]
    ref = svi.BundleAdjuster(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"])
    synth.build_ba_graph(ref, prob)
    ref.initialize()
    done = list(ref.optimize_until())
    T, p, chi = ref.get_poses()[1], ref.get_landmarks()[1], ref.chi2()
    ref.close()
    assert done[1] >= 21
    out = _run_sharded(svi, prob, 8, (), until=True)
    for o in out:
        assert o[0] == done and o[6] == 0
        assert o[7] == out[0][7] and np.array_equal(o[1], out[0][1])
        assert np.abs(o[1] - T).max() < 1e-8 * max(1.0, np.abs(T).max()) and np.abs(o[2] - p).max() < 1e-8 * np.abs(p).max()
        assert abs(o[3][0] - chi[0]) <= 1e-9 * chi[0]


def test_more_ranks_than_work(svi):
    """Eight ranks on a graph whose landmarks do not fill them evenly: shards are cut by edge count, so a rank may own very few
    landmarks or none at all - it still takes part in every collective and ends with the same estimates."""
    prob = synth.make_ba_problem(8, 12, 60, seed=11)
    cam = prob["cam"]
    iters = (1, 4)
    ref = svi.BundleAdjuster(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"])
    synth.build_ba_graph(ref, prob)
    ref.initialize()
    done = [ref.optimize(n) for n in iters]
    T, p = ref.get_poses()[1], ref.get_landmarks()[1]
    out = _run_sharded(svi, prob, 8, iters)
    assert sum(o[4] for o in out) == prob["n_lm"]
    for o in out:
        assert o[0] == done
        assert np.abs(o[1] - T).max() < 1e-9 and np.abs(o[2] - p).max() < 1e-9 * max(1.0, np.abs(p).max())
        assert np.array_equal(o[1], out[0][1])


def test_failed_trial_is_failed_on_every_rank(svi):
    """one landmark block is not positive definite until lambda has grown (a prior with negative information): only the
    rank that owns the landmark sees that in its status word - the failure has to travel with the reduced scalars, or the
    ranks would take different branches of the LM rule"""
    prob = synth.make_ba_problem(12, 400, 2600, seed=5)
    cam = prob["cam"]

    def extra(ba):
        ba.add_landmark(900000, prob["lm_init"][3], fixed=True)
        ba.add_edge_lm_lm(3, 900000, np.zeros(3), [-1e4, 0, 0, -1e4, 0, -1e4], robust=False)

    iters = (1, 3)
    ref = svi.BundleAdjuster(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"])
    synth.build_ba_graph(ref, prob)
    extra(ref)
    ref.initialize()
    done = [ref.optimize(n) for n in iters]
    st = ref.stats()
    assert st.chol_failures > 0, "the graph was supposed to provoke failed trials"
    out = _run_sharded(svi, prob, 2, iters, extra)
    for o in out:
        assert o[0] == done
        assert o[5] == st.lm_trials and o[6] == st.chol_failures
        assert np.abs(o[1] - ref.get_poses()[1]).max() < 1e-8


@pytest.mark.parametrize("rot,seed", [(0.3, 3), (0.35, 6), (0.35, 8)])
def test_rejected_trials_identical_on_every_rank(svi, rot, seed):
    """Overshooting Gauss-Newton steps (the graphs of test_ba_gpu.test_rejected_trials_follow_g2o): 0 < rho < 1 and rejected
    trials, so lambda depends on the step scale sum dx (lambda dx + b).  In the iterations whose linearisation stays local
    every rank holds only its partial b_p: the pose part of the scale must still come out as one global sum, or the ranks
    damp differently and drift apart (round-1 advisor finding)."""
    from test_ba_gpu import _nonlinear_graph
    iters = (1, 3, 3)
    ref = _nonlinear_graph(svi.BundleAdjuster, 1e-12, rot, seed=seed)
    ref.initialize()
    done, lams = [], []
    for n in iters:
        done.append(ref.optimize(n))
        lams.append(ref.lm_lambda)
    st = ref.stats()
    assert st.lm_trials > st.lm_iterations, "the graph was supposed to provoke rejected trials"

    def make(rank, n_ranks):
        import functools
        cls = functools.partial(svi.BundleAdjuster, rank=rank, n_ranks=n_ranks)
        return _nonlinear_graph(cls, 1e-12, rot, seed=seed)

    out = _run_sharded(svi, None, 2, iters, make=make)
    for o in out:
        assert o[0] == done and o[5] == st.lm_trials
        assert np.allclose(o[7], lams, rtol=1e-4, atol=0)       # the same damping history as one rank ...
        assert o[7] == out[0][7]                                # ... and bit-identical between the ranks
        assert np.array_equal(o[1], out[0][1])
        assert np.abs(o[1] - ref.get_poses()[1]).max() <= 1e-4 * np.abs(ref.get_poses()[1]).max()


def test_missing_hook_fails_loudly(svi):
    prob = synth.make_ba_problem(6, 60, 330, seed=42)
    cam = prob["cam"]
    ba = svi.BundleAdjuster(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"], rank=0, n_ranks=2)
    synth.build_ba_graph(ba, prob)
    ba.initialize()
    with pytest.raises(svi.SviError) as e:
        ba.optimize(1)
    assert e.value.status == 4


def test_rccl_hook_aliases_library_memory(svi):
    """svi_mapper_amd.dist.make_allreduce_hook on a 1-rank RCCL group: the hook must wrap the raw device
    pointer without copying and run the collective on the given stream (sum over 1 rank = identity)."""
    import os
    import socket

    import torch
    import torch.distributed as dist

    from svi_mapper_amd import dist as sdist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        hook = sdist.make_allreduce_hook()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            x = torch.arange(1000, dtype=torch.float64, device="cuda")
        assert hook(x.data_ptr(), x.numel(), st.cuda_stream) == 0
        with torch.cuda.stream(st):
            x += 1.0  # ordered after the collective on the same stream
        st.synchronize()
        assert torch.equal(x.cpu(), torch.arange(1000, dtype=torch.float64) + 1.0)
    finally:
        dist.destroy_process_group()


def test_shard_driven_through_rccl(svi):
    """Shard 0 of 2 driven through a real RCCL communicator (1 rank: every sum is the identity, so the handle solves
    the sub-problem of its own landmarks). Checks the collective schedule: the pose sums are exchanged on their own only
    in front of the first trial of a block (lambda_0 needs max |H_jj|); otherwise they ride with the reduced system -
    two collectives per trial (reduced system, four scalars), all on the handle's stream."""
    import os
    import socket

    import torch
    import torch.distributed as dist

    from svi_mapper_amd import dist as sdist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    prob = synth.make_ba_problem(40, 3000, 20000, seed=5)
    cam = prob["cam"]
    try:
        calls = []
        inner = sdist.make_allreduce_hook()

        def hook(ptr, count, stream):
            calls.append((count, stream))
            return inner(ptr, count, stream)

        ba = svi.BundleAdjuster(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"], rank=0, n_ranks=2)
        synth.build_ba_graph(ba, prob)
        ba.set_allreduce(hook)
        ba.initialize()
        blocks = (1, 6)
        done = [ba.optimize(n) for n in blocks]
        st = ba.stats()
        chi = ba.chi2()
        n_before = len(calls)
        ba.get_landmarks()
        ba.get_poses()
        assert len(calls) == n_before + 1
        ba.close()
    finally:
        dist.destroy_process_group()
    assert done == list(blocks) and np.isfinite(chi[0])
    assert len({stream for _, stream in calls}) == 1
    scalars = [c for c, _ in calls if c == 4]
    downloads = [c for c, _ in calls if c == 3 * 3000]  # the gather of the landmark shards (3000 landmarks)
    big = [c for c, _ in calls if c != 4 and c != 3 * 3000]
    assert len(downloads) == 1  # lazily, by the first call that reads the estimates (none between the blocks)
    assert len(scalars) == st.lm_trials
    assert len(big) == st.lm_trials + len(blocks), (len(big), st.lm_trials)
    # the reduced system carries two more doubles when the pose sums ride along
    assert len(set(big)) == 3 and max(big) - sorted(set(big))[-2] == 2, sorted(set(big))


def test_dlpack_alias_on_device(svi):
    """the hooks and the extractor callbacks hand library-owned HBM to torch through a DLPack capsule that states the device
    (svi_mapper_amd/_dlpack.py): the tensor aliases the memory, on the device it was told, for every dtype the library uses"""
    import torch
    from svi_mapper_amd import _dlpack
    for dtype in (torch.float64, torch.float32, torch.int32, torch.uint8):
        src = (torch.arange(24, device="cuda:0") % 7).to(dtype).reshape(4, 6).contiguous()
        t = _dlpack.alias(src.data_ptr(), (4, 6), dtype, torch.device("cuda", 0))
        assert t.device == src.device and t.dtype == dtype and t.data_ptr() == src.data_ptr()
        assert torch.equal(t, src)
        t[2, 3] = 5
        assert src[2, 3].item() == 5
    assert _dlpack.alias(0, (0, 4), torch.float32, torch.device("cuda", 0)).shape == (0, 4)
