"""dev helper: where the host time of an eager Renderer step goes (no GPU syncs inside the timed parts)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unicorn_amd import native, synth
from unicorn_amd.renderer import Renderer
from unicorn_amd.structures import Meshes, TexturesUV
dev = torch.device("cuda", 0)
B, H, K = 32, 128, 50
sc = synth.make_scene(B, level=3, tex_size=64, seed=0, device=dev)
rend = Renderer(H, cameras={"name": "fov", "fov": 30}, faces_per_pixel=K, sigma=1e-4).to(dev)
verts = sc["verts"].requires_grad_(True); maps = sc["maps"].requires_grad_(True)
R = sc["R"].requires_grad_(True); T = sc["T"].requires_grad_(True)
fb, ub = sc["faces"][None].expand(B, -1, -1), sc["verts_uvs"][None].expand(B, -1, -1)
cot = torch.randn(B, 4, H, H, device=dev)
acc = {"zero": 0.0, "meshes": 0.0, "forward": 0.0, "backward": 0.0}
def step(rec=False):
    t0 = time.perf_counter()
    for t in (verts, maps, R, T): t.grad = None
    t1 = time.perf_counter()
    meshes = Meshes(verts=verts, faces=fb, textures=TexturesUV(maps, fb, ub))
    t2 = time.perf_counter()
    img = rend(meshes, R, T)
    t3 = time.perf_counter()
    img.backward(cot)
    t4 = time.perf_counter()
    if rec:
        acc["zero"] += t1 - t0; acc["meshes"] += t2 - t1; acc["forward"] += t3 - t2; acc["backward"] += t4 - t3
for _ in range(50): step()
torch.cuda.synchronize()
n = 500
for rep in range(3):
    for k in acc: acc[k] = 0.0
    t0 = time.perf_counter()
    for _ in range(n): step(True)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("host per step %.1f us (wall %.1f) :" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6), {k: round(v / n * 1e6, 1) for k, v in acc.items()})
# the two C calls alone (same arguments every time), and an empty autograd round trip
import torch.autograd as ag
class Nop(ag.Function):
    @staticmethod
    def forward(ctx, a, b, c, d): return a.new_empty(1)
    @staticmethod
    def backward(ctx, g): return None, None, None, None
out = None
t0 = time.perf_counter()
for _ in range(n):
    o = Nop.apply(verts, maps, R, T); o.backward(o.detach())
t1 = time.perf_counter()
print("empty Function fwd+bwd round trip: %.1f us" % ((t1 - t0) / n * 1e6))
# the native wrappers alone (allocations + argument marshalling + the C call), then the bare C call cost by difference
import math
s = 1.0 / math.tan(math.radians(15.0))
F = sc["faces"].shape[0]
first = torch.arange(B, device=dev, dtype=torch.int64) * F
num = torch.full((B,), F, device=dev, dtype=torch.int64)
sigma = 1e-4
blur = math.log(1.0 / 1e-4 - 1.0) * sigma
fuv = sc["verts_uvs"][sc["faces"]].contiguous()
vd, md, Rd, Td = verts.detach(), maps.detach(), R.detach(), T.detach()
def nf():
    return native.render_meshes_forward(vd, sc["faces"], Rd, Td, s, s, first, num, (H, H), blur, K, True, md, fuv, True,
                                        sigma, (1.0, 1.0, 1.0), None, 1, True, True)
def nb(st, gm):
    return native.render_meshes_backward(cot, vd, sc["faces"], Rd, Td, s, s, first, num, (H, H), blur, K, True, md, fuv,
                                         True, sigma, (1.0, 1.0, 1.0), None, 1, st, gm, True, True, True, True, True)
for _ in range(20):
    img, st, gm = nf(); nb(st, gm)
torch.cuda.synchronize()
tf = tb = 0.0
for _ in range(n):
    t0 = time.perf_counter(); img, st, gm = nf(); t1 = time.perf_counter(); nb(st, gm); t2 = time.perf_counter()
    tf += t1 - t0; tb += t2 - t1
torch.cuda.synchronize()
print("native wrappers alone: forward %.1f us, backward %.1f us" % (tf / n * 1e6, tb / n * 1e6))
t0 = time.perf_counter()
for _ in range(n):
    a = torch.empty(B, 4, H, H, dtype=torch.float32, device=dev)
t1 = time.perf_counter()
print("one torch.empty: %.2f us" % ((t1 - t0) / n * 1e6))
