"""stereo posit: ms per solve (500 points) as bench.py's frontend block measures it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import svi_mapper_amd as svi
from svi_mapper_amd import temporal
import posit_case, track_scene as ts
dev = torch.device("cuda", 0)
d = lambda a: torch.tensor(np.ascontiguousarray(a), device=dev)
for n in (500, 2000, 500, 2000):
    c = posit_case.make(n, 2)
    solver = temporal.SolverStereoPosit(ts.P_LEFT, ts.P_RIGHT, device=0)
    px, pl, pr = d(c["xyz"]), d(c["uvl"]), d(c["uvr"])
    r = solver.solve(c["T_last"], c["t_imu"], c["T_est"], px, pl, pr)
    for _ in range(5): solver.solve(c["T_last"], c["t_imu"], c["T_est"], px, pl, pr)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): solver.solve(c["T_last"], c["t_imu"], c["T_est"], px, pl, pr)
    torch.cuda.synchronize()
    print("n %d: %.1f us per solve, %d iterations, status %d" % (n, 1e6 * (time.perf_counter() - t0) / 200, r.iterations, r.status), flush=True)
