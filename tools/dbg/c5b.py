import sys, numpy as np, torch
sys.path.insert(0, '.')
from svi_mapper_amd import vi_stream
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 25
s = vi_stream.ViStream(N, dev, step=0.08)
trk = vi_stream.OnlineTracker(s)
trk.start(s.render(0))
print("start", trk.n_used)
for t in range(1, N):
    before = dict(trk.stats)
    v = trk.step(t, s.render(t))
    st = trk.stats
    if t % 10 == 0 or st["ba_calls"] != before["ba_calls"] or not np.isfinite(trk.T_w2l).all(): print(t, "vis", v, "s1", st["stage1"] - before["stage1"], "s2", st["stage2"] - before["stage2"], "s3", st["stage3"] - before["stage3"],
          "pf", st["posit_fail"], "err", ["%.4f" % x for x in trk.pose_error(t)], "n", trk.n_used, "kf", len(trk.key_frames), "ba", st["ba_calls"])
    if not np.isfinite(trk.T_w2l).all():
        break
