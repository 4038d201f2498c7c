"""(round-2 tool: its stand-ins have the round-2 signatures of the native wrappers)
dev helper (no GPU needed): the eager step's Python glue alone -- the two native wrappers replaced by stand-ins that
allocate what they would -- timed and profiled on CPU tensors."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unicorn_amd import native, ops, synth
from unicorn_amd.renderer import Renderer
from unicorn_amd.structures import Meshes, TexturesUV
B, H, K = 32, 128, 50
sc = synth.make_scene(B, level=3, tex_size=64, seed=0, device="cpu")
rend = Renderer(H, cameras={"name": "fov", "fov": 30}, faces_per_pixel=K, sigma=1e-4)
verts = sc["verts"].requires_grad_(True); maps = sc["maps"].requires_grad_(True)
R = sc["R"].requires_grad_(True); T = sc["T"].requires_grad_(True)
fb, ub = sc["faces"][None].expand(B, -1, -1), sc["verts_uvs"][None].expand(B, -1, -1)
cot = torch.randn(B, 4, H, H)
def fake_fwd(verts, faces, R, T, sx, sy, first, nf, image_size, blur, K, clip, maps, fuv, shared, sigma, bg, amb, mode,
             need_geometry_grads=False, need_map_grads=False, cull_backfaces=False):
    return torch.empty(verts.shape[0], 4, *image_size), torch.empty(16, dtype=torch.uint8), torch.empty_like(maps)
def fake_bwd(g, verts, faces, R, T, *a):
    return torch.empty_like(verts), torch.empty_like(R), torch.empty_like(T), a[-6]
native.render_meshes_forward = fake_fwd
native.render_meshes_backward = fake_bwd
def step():
    for t in (verts, maps, R, T): t.grad = None
    meshes = Meshes(verts=verts, faces=fb, textures=TexturesUV(maps, fb, ub))
    img = rend(meshes, R, T)
    img.backward(cot)
for _ in range(200): step()
n = 3000
t0 = time.perf_counter()
for _ in range(n): step()
print("glue per step: %.1f us" % ((time.perf_counter() - t0) / n * 1e6))
if len(sys.argv) > 1:
    pr = cProfile.Profile(); pr.enable()
    for _ in range(n): step()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
