"""dev helper: time of the visualisation path (hard raster at 2x the viz size, K=1, 2x2 average) and of
compute_vertex_visibility, per call."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unicorn_amd import synth
from unicorn_amd.renderer import Renderer
from unicorn_amd.structures import Meshes, TexturesUV
dev = torch.device("cuda", 0)
for B, level in ((32, 3), (8, 3), (8, 3), (32, 3), (8, 5), (1, 3)):
    sc = synth.make_scene(B, level=level, tex_size=64, seed=0, device=dev)
    rend = Renderer(64, cameras={"name": "fov", "fov": 30}, faces_per_pixel=25, sigma=1e-4).to(dev)
    fb, ub = sc["faces"][None].expand(B, -1, -1), sc["verts_uvs"][None].expand(B, -1, -1)
    meshes = Meshes(verts=sc["verts"], faces=fb, textures=TexturesUV(sc["maps"], fb, ub))
    for name, fn in (("viz render", lambda: rend(meshes, sc["R"], sc["T"], viz_purpose=True)),
                     ("vertex visibility", lambda: rend.compute_vertex_visibility(meshes, sc["R"], sc["T"]))):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): fn()
        torch.cuda.synchronize()
        print("B=%d level %d %s: %.3f ms per call" % (B, level, name, (time.perf_counter() - t0) / 20 * 1e3))
