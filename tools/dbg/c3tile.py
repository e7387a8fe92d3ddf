import sys, time
sys.path.insert(0, '.')
import torch, bench, svi_mapper_amd as svi
from svi_mapper_amd import synth
for name, prob in (("c3", synth.make_c3()), ("c4", bench.cached_problem(1)), ("vi", synth.make_vi_problem(120, 12000, 80000))):
    cam = prob["cam"]
    for tile in (48, 96):
        ba = svi.BundleAdjuster(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"], chol_tile=tile)
        synth.build_ba_graph(ba, prob)
        ba.initialize()
        bench.run_exact(ba, 5)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        bench.run_exact(ba, 20)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        st = ba.stats()
        print(name, "tile", tile, "levels", st.chol_steps, "tiles", st.chol_tiles_nnz, "ms/it %.4f" % (1e3 * dt / 20), flush=True)
        ba.close()
