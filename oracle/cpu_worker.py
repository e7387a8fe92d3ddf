"""One independent CPU solve of the benchmark graph with the oracle (bench.py's all-cores cpu_baseline variant starts one
of these per core, as child processes that never touch the GPU).  TEST INFRASTRUCTURE like everything under oracle/.

    python oracle/cpu_worker.py <problem.npz> <iterations> <liboracle path>

prints one JSON line {"iterations": n, "seconds": t} (graph construction excluded, like the single-thread leg)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[0] = ROOT   # (the script's own directory would shadow the package `oracle` with the module oracle/oracle.py)


def main():
    path, iters, libpath = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    from oracle import oracle as orc
    from svi_mapper_amd import synth
    z = np.load(path)
    keys = ("R_true", "t_true", "R_init", "t_init", "lm_true", "lm_init", "obs_kf", "obs_lm", "uvL", "uvR", "xyz")
    prob = {k: z[k] for k in keys}
    prob.update(cam=synth.kitti_camera(), n_kf=int(z["n_kf"]), n_lm=int(z["n_lm"]))
    cam = prob["cam"]
    o = orc.OracleBA(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"], lib=orc.load(libpath))
    synth.build_ba_graph(o, prob)
    o.initialize()
    print("ready", flush=True)
    sys.stdin.readline()          # all workers start their clocks together
    t0 = time.time()
    done = 0
    while done < iters:
        done += o.optimize(iters - done)
    print(json.dumps({"iterations": done, "seconds": time.time() - t0}), flush=True)


if __name__ == "__main__":
    main()
