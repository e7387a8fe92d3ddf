import sys, time, cProfile, pstats
sys.path.insert(0, '.')
import torch
from svi_mapper_amd import vi_stream
dev = torch.device("cuda", 0)
N = 300
s = vi_stream.ViStream(N, dev, step=0.08)
frames = [s.render(t) for t in range(N)]
trk = vi_stream.OnlineTracker(s)
trk.start(frames[0])
for t in range(1, 100):
    trk.step(t, frames[t])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
t0 = time.perf_counter()
for t in range(100, N):
    trk.step(t, frames[t])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
pr.disable()
print("ms/frame", 1e3 * dt / (N - 100), "n_used", trk.n_used)
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
