"""dev probe: one B=32 render step (forward + backward) against the same batch split into S sub-batches on S streams
(each through its own forward -> backward chain), all replayed as one hipGraph.  usage: stream_split_probe.py [S ...]"""
import math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unicorn_amd import synth
from unicorn_amd.graphs import GraphedStep
from unicorn_amd.renderer import Renderer
from unicorn_amd.structures import Meshes, TexturesUV

B, H, K, LEVEL = [int(x) for x in os.environ.get('UR_DEV_SHAPE', '32,128,50,3').split(',')]
dev = torch.device("cuda", 0)
sc = synth.make_scene(B, level=LEVEL, tex_size=64, seed=0, device=dev)
rend = Renderer(H, cameras={"name": "fov", "fov": 30}, faces_per_pixel=K, sigma=1e-4).to(dev)
cot = torch.randn(B, 4, H, H, device=dev)


def build(S):
    parts = []
    for idx in torch.arange(B, device=dev).chunk(S):
        lo, hi = int(idx[0]), int(idx[-1]) + 1
        n = hi - lo
        leaves = [sc[k][lo:hi].clone().requires_grad_(True) for k in ("verts", "maps", "R", "T")]
        fb, ub = sc["faces"][None].expand(n, -1, -1), sc["verts_uvs"][None].expand(n, -1, -1)
        parts.append((leaves, fb, ub, cot[lo:hi].contiguous(), torch.cuda.Stream() if lo else None))
    def fn():
        cur = torch.cuda.current_stream()
        for leaves, fb, ub, c, st in parts:
            if st is not None:
                st.wait_stream(cur)
            with torch.cuda.stream(st if st is not None else cur):
                v, m, R, T = leaves
                img = rend(Meshes(verts=v, faces=fb, textures=TexturesUV(m, fb, ub)), R, T)
                img.backward(c)
        for _, _, _, _, st in parts:
            if st is not None:
                cur.wait_stream(st)
    return GraphedStep(fn, grads_of=[t for p in parts for t in p[0]])


for S in [int(x) for x in sys.argv[1:]] or [1, 2, 4]:
    g = build(S)
    for _ in range(20): g()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 200
    for _ in range(n): g()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("streams %d : %.1f us/step, %.0f images/s" % (S, dt * 1e6, B / dt), flush=True)
