"""dev helper: depth statistics of the headline workload (how many (pixel, layer) items the fused kernels process)."""
import math, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unicorn_amd import native, synth, ops
from unicorn_amd.renderer import Renderer

B, H, K = 32, 128, 50
dev = torch.device("cuda", 0)
sc = synth.make_scene(B, level=3, tex_size=64, seed=0, device=dev)
s = 1.0 / math.tan(math.radians(15.0))
F = sc["faces"].shape[0]
fv = native.project_faces(sc["verts"], sc["faces"], sc["R"], sc["T"], s, s)
first = torch.arange(B, device=dev, dtype=torch.int64) * F
num = torch.full((B,), F, device=dev, dtype=torch.int64)
sigma = 1e-4
blur = math.log(1.0 / 1e-4 - 1.0) * sigma
fuv = sc["verts_uvs"][sc["faces"]].contiguous()
img, layers, ws = native.render_layered_forward(fv, first, num, (H, H), blur, K, True, sc["maps"], fuv, True, sigma,
                                                (1, 1, 1), None, 1)
torch.cuda.synchronize()
L = layers.view(B, H // 8, H // 8, K, 6, 64)[:, :, :, :, 0, :]       # fid planes
alive = (L >= 0).to(torch.int32).cumprod(dim=3)                        # leading run
m = alive.sum(dim=3)                                                   # (B,TH,TW,64) layers per pixel
print("items", int(m.sum()), "per image", float(m.sum()) / B, "covered px frac", float((m > 0).float().mean()))
print("mean layers on covered px", float(m[m > 0].float().mean()), "max", int(m.max()))
t = m.sum(dim=3)                                                       # per tile
print("tile items: mean", float(t.float().mean()), "max", int(t.max()), "p99", float(t.float().flatten().quantile(0.99)))
tm = m.max(dim=3).values
print("tile max depth: mean", float(tm.float().mean()), "max", int(tm.max()))
print("lockstep cost sum(tile maxdepth*64)", int(tm.sum()) * 64, "vs items", int(m.sum()))
hist = torch.bincount(m.flatten(), minlength=16)[:24]
print("hist layers/pixel", hist.tolist())
bins = t.view(B, H // 16, 2, H // 16, 2).sum(dim=(2, 4))
print("bin items: mean", float(bins.float().mean()), "max", int(bins.max()), "nonempty", int((bins > 0).sum()), "of", bins.numel())
if os.environ.get("UR_DEBUG") == "13":
    cot = torch.randn(B, 4, H, H, device=dev)
    gfv, gm = native.render_layered_backward(cot, first, num, B * F, (H, H), blur, K, True, sc["maps"], fuv, True,
                                             sigma, (1, 1, 1), None, 1, layers, ws)
    torch.cuda.synchronize()
    print("texel claims: fail", float(gm.flatten()[0]), "ok", float(gm.flatten()[1]))
    print("face claims: fail", float(gfv.flatten()[0]), "ok", float(gfv.flatten()[1]))
