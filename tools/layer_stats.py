"""dev helper: depth statistics of the headline workload (how many (pixel, layer) items the fused kernels process)."""
import math, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unicorn_amd import native, synth, ops
from unicorn_amd.renderer import Renderer

B, H, K, LEVEL = [int(x) for x in os.environ.get('UR_DEV_SHAPE', '32,128,50,3').split(',')]  # batch, image, K, icosphere level
dev = torch.device("cuda", 0)
sc = synth.make_scene(B, level=LEVEL, tex_size=64, seed=0, device=dev)
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
nb = (H // 16) ** 2
cnt_off = B * nb * 4 * K * 6 * 64                                        # per-pixel layer counts follow the records
m = layers.view(-1)[cnt_off:cnt_off + B * nb * 4 * 64].view(B, nb, 4, 64)  # (image, bin, wave group, pixel)
print("items", int(m.sum()), "per image", float(m.sum()) / B, "covered px frac", float((m > 0).float().mean()))
print("mean layers on covered px", float(m[m > 0].float().mean()), "max", int(m.max()))
t = m.sum(dim=3)                                                       # per wave group
print("group items: mean", float(t.float().mean()), "max", int(t.max()), "p99", float(t.float().flatten().quantile(0.99)))
tm = m.max(dim=3).values
print("group max depth: mean", float(tm.float().mean()), "max", int(tm.max()))
h = torch.bincount(m.flatten().clamp(max=20), minlength=21)
print("depth histogram (0..20+):", h.tolist())
