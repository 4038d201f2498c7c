"""dev helper: host time of an eager step by layer (perf_counter stamps around the layers' entry points)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unicorn_amd import native, synth, ops
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
acc = {}
def wrap(obj, name, tag):
    fn = getattr(obj, name)
    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[tag] = acc.get(tag, 0.0) + time.perf_counter() - t0
    setattr(obj, name, w)
wrap(native, "render_meshes_forward", "native fwd")
wrap(native, "render_meshes_backward", "native bwd")
f0, b0 = ops._RenderMeshes.forward, ops._RenderMeshes.backward
def fwd(ctx, *a):
    t0 = time.perf_counter(); r = f0(ctx, *a); acc["Function.forward body"] = acc.get("Function.forward body", 0.0) + time.perf_counter() - t0; return r
def bwd(ctx, g):
    t0 = time.perf_counter(); r = b0(ctx, g); acc["Function.backward body"] = acc.get("Function.backward body", 0.0) + time.perf_counter() - t0; return r
ops._RenderMeshes.forward = staticmethod(fwd)
ops._RenderMeshes.backward = staticmethod(bwd)
wrap(ops, "render_meshes", "ops.render_meshes (apply)")
def step(rec=False):
    t0 = time.perf_counter()
    for t in (verts, maps, R, T): t.grad = None
    meshes = Meshes(verts=verts, faces=fb, textures=TexturesUV(maps, fb, ub))
    t2 = time.perf_counter()
    img = rend(meshes, R, T)
    t3 = time.perf_counter()
    img.backward(cot)
    t4 = time.perf_counter()
    if rec:
        acc["Renderer.__call__"] = acc.get("Renderer.__call__", 0.0) + t3 - t2
        acc["Tensor.backward"] = acc.get("Tensor.backward", 0.0) + t4 - t3
for _ in range(50): step()
torch.cuda.synchronize()
n = 1000
for rep in range(2):
    acc.clear()
    for _ in range(n): step(True)
    torch.cuda.synchronize()
    print({k: round(v / n * 1e6, 1) for k, v in sorted(acc.items())})
