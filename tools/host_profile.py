"""dev helper: cProfile of the eager Renderer step (host side only; the GPU runs behind)."""
import cProfile, os, pstats, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unicorn_amd import synth
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
def step():
    for t in (verts, maps, R, T): t.grad = None
    meshes = Meshes(verts=verts, faces=fb, textures=TexturesUV(maps, fb, ub))
    img = rend(meshes, R, T)
    img.backward(cot)
for _ in range(50): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(1000): step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(45)
