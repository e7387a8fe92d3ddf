import sys, numpy as np, torch
sys.path.insert(0, '.')
from svi_mapper_amd import vi_stream, temporal
dev = torch.device("cuda", 0)
s = vi_stream.ViStream(40, dev, step=0.1)
L, R = s.render(0)
print("img", L.shape, L.float().mean().item(), L.float().std().item(), (L == 128).float().mean().item())
trk = vi_stream.OnlineTracker(s)
trk.brief.set_image("left", L); trk.brief.set_image("right", R)
T_w2l = vi_stream.inv12(s.T_l2w[0])
want = 650
g = int(np.ceil(np.sqrt(want * 1.6)))
uu = np.linspace(60, vi_stream.W - 60, g)[None, :].repeat(g, 0)
vv = np.linspace(40, vi_stream.H - 40, g)[:, None].repeat(g, 1)
uv = np.rint(np.stack([uu.ravel(), vv.ravel()], 1))
uv_d = torch.tensor(uv, device=dev)
P, lam = s.ground_point(s.T_l2w[0], uv_d)
print("lam", lam.min().item(), lam.max().item(), ((lam > 0.5) & (lam < 25)).sum().item(), len(uv))
good = (lam > 0.5) & (lam < 25.0)
uv_d = uv_d[good]
n = uv_d.shape[0]
roi = torch.tensor([[0.0, 0.0, float(vi_stream.W), float(vi_stream.H)]], dtype=torch.float32, device=dev)
seg = torch.tensor([0, n], dtype=torch.int32, device=dev)
seg_o, kp_o, desc = trk.brief("left", roi, seg, uv_d.float().contiguous())
print("brief kept", kp_o.shape, n)
kp = torch.full((n,), 7.0, dtype=torch.float32, device=dev)
res = trk.fm.add_new_landmarks(trk.brief, uv_d.float().contiguous(), kp, desc.contiguous())
print("status hist", np.bincount(res.status.cpu().numpy(), minlength=9))
ok = res.status == 0
print("disp", (res.uv_left[ok, 0] - res.uv_right[ok, 0])[:10], "z", res.xyz_left[ok, 2][:10], lam[good][ok][:10])
