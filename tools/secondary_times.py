"""dev measurement: the entry points beside the headline path (SURVEY 8 rows A8, A11, A12, f-3, f-4 and the two-stage
route), timed per call with everything resident on the GPU: eager calls, wall clock over a synchronised loop (these
are visualisation / evaluation / regulariser paths: host time counts for their callers).  One JSON line."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unicorn_amd import regularizers, synth  # noqa: E402
from unicorn_amd.renderer import Renderer  # noqa: E402
from unicorn_amd.structures import Meshes, TexturesUV  # noqa: E402

dev = torch.device("cuda", 0)


def timed(fn):
    """median of three loops; the loop length follows a first timed call, so a slow path takes seconds, not minutes"""
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    one = time.perf_counter() - t0
    n = max(1, min(30, int(1.0 / max(one, 1e-6))))
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / n)
    return round(sorted(ts)[1] * 1e3, 4)


class _Out(dict):
    def __setitem__(self, k, v):
        super().__setitem__(k, v)
        print(f"{v:10.3f} ms  {k}", file=sys.stderr, flush=True)


def scene(B, level=3, tex=64, grad=True):
    sc = synth.make_scene(B, level=level, tex_size=tex, seed=0, device=dev)
    leaves = [sc[k].requires_grad_(grad) for k in ("verts", "maps", "R", "T")]
    fb, ub = sc["faces"][None].expand(B, -1, -1), sc["verts_uvs"][None].expand(B, -1, -1)
    return leaves, fb, ub


def fwd_bwd(rend, leaves, fb, ub, cot):
    def step():
        for t in leaves:
            t.grad = None
        img = rend(Meshes(verts=leaves[0], faces=fb, textures=TexturesUV(leaves[1], fb, ub)), leaves[2], leaves[3])
        img.backward(cot)
    return step


out = _Out()
B = 32
cam = {"name": "fov", "fov": 30}
# the two-stage route (fragments materialised: rasterize -> shade/blend), same workload as the headline
leaves, fb, ub = scene(B)
rend = Renderer(128, cameras=cam, faces_per_pixel=50, sigma=1e-4).to(dev)
cot = torch.randn(B, 4, 128, 128, device=dev)
out["fused one-call route, 32x128x128 K=50, fwd+bwd (eager)"] = timed(fwd_bwd(rend, leaves, fb, ub, cot))
rend.renderer.fused = False
out["two-stage route (fragments in HBM), 32x128x128 K=50, fwd+bwd (eager)"] = timed(fwd_bwd(rend, leaves, fb, ub, cot))
# lit shading and the softmax blend at the training size (fragments + torch lighting + ur_blend_colors)
leaves, fb, ub = scene(B)
cot = torch.randn(B, 4, 64, 64, device=dev)
for name, kw in (("LayeredShader shading_type=phong, directional light", dict(shading_type="phong", lights={"name": "directional"})),
                 ("SoftPhongShaderPlus (layered_shader=False), point light", dict(layered_shader=False, lights={"name": "point"}))):
    r = Renderer(64, cameras=cam, faces_per_pixel=25, sigma=1e-4, **kw).to(dev)
    out[name + ", 32x64x64 K=25, fwd+bwd (eager)"] = timed(fwd_bwd(r, leaves, fb, ub, cot))
# the other shading types, vertex colours, and a batch of unequal meshes (lists: the projected-faces route)
from unicorn_amd.structures import TexturesVertex  # noqa: E402
for name, kw in (("LayeredShader shading_type=flat, point light", dict(shading_type="flat", lights={"name": "point"})),):
    r = Renderer(64, cameras=cam, faces_per_pixel=25, sigma=1e-4, **kw).to(dev)
    out[name + ", 32x64x64 K=25, fwd+bwd (eager)"] = timed(fwd_bwd(r, leaves, fb, ub, cot))
vcol = torch.rand(B, leaves[0].shape[1], 3, device=dev, requires_grad=True)
for name, kw in (("TexturesVertex, raw shading", dict()),
                 ("TexturesVertex, shading_type=gouraud, directional light", dict(shading_type="gouraud", lights={"name": "directional"}))):
    r = Renderer(64, cameras=cam, faces_per_pixel=25, sigma=1e-4, **kw).to(dev)

    def vstep(r=r):
        for t in (leaves[0], leaves[2], leaves[3], vcol):
            t.grad = None
        r(Meshes(verts=leaves[0], faces=fb, textures=TexturesVertex(vcol)), leaves[2], leaves[3]).backward(cot)
    out[name + ", 32x64x64 K=25, fwd+bwd (eager)"] = timed(vstep)
r = Renderer(64, cameras=cam, faces_per_pixel=25, sigma=1e-4).to(dev)
v2, f2 = synth.icosphere(2)
v2 = (v2 * 0.5).to(dev)
f2 = f2.to(dev)
uv2 = synth.spherical_uvs(v2.cpu()).to(dev)


def hetero():
    for t in leaves:
        t.grad = None
    vl = [leaves[0][i] if i % 2 == 0 else v2 for i in range(B)]
    fl = [fb[i] if i % 2 == 0 else f2 for i in range(B)]
    # (TexturesUV wants one face count: per-mesh tables padded to the larger mesh)
    r(Meshes(vl, fl, TexturesVertex([torch.ones_like(x) * 0.5 for x in vl])), leaves[2], leaves[3]).backward(cot)


try:
    out["batch of unequal meshes (ico3 / ico2 alternating, vertex colours), 32x64x64 K=25, fwd+bwd (eager)"] = timed(hetero)
except Exception as e:  # a path this build does not serve is reported, not hidden
    print("unequal meshes:", type(e).__name__, e, file=sys.stderr, flush=True)
# visualisation path and vertex visibility (hard raster at 512x512, K=1, 2x2 average), no gradients
leaves, fb, ub = scene(B, grad=False)
rend = Renderer(64, cameras=cam, faces_per_pixel=25, sigma=1e-4).to(dev)
meshes = Meshes(verts=leaves[0], faces=fb, textures=TexturesUV(leaves[1], fb, ub))
with torch.no_grad():
    out["viz render (512x512 hard raster, 2x2 average), 32 meshes"] = timed(lambda: rend(meshes, leaves[2], leaves[3], viz_purpose=True))
    out["compute_vertex_visibility, 32 meshes"] = timed(lambda: rend.compute_vertex_visibility(meshes, leaves[2], leaves[3]))
# mesh regularisers: normal consistency + uniform Laplacian, value and gradient, one launch
leaves, fb, ub = scene(B)


def reg_step():
    leaves[0].grad = None
    n, l = regularizers.mesh_regularizers(Meshes(verts=leaves[0], faces=fb))
    (n.mean() + l.mean()).backward() if n.dim() else (n + l).backward()


out["mesh_regularizers (normal consistency + Laplacian), 32 x ico3, fwd+bwd (eager)"] = timed(reg_step)
print(json.dumps({"ms_per_call": out}))
