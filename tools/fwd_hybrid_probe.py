"""dev helper (timing only, forward only): the fused forward at the headline size with the heaviest UR_DEV_HYBRID bins run
as eight-wave workgroups on a second stream beside the four-wave launch of the rest (variant built from abx/hyb.hip).
Never run a backward on this variant's layers: the two launches leave them in two layouts."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unicorn_amd import native, synth  # noqa: E402

B, H, K, LEVEL = [int(x) for x in os.environ.get("UR_DEV_SHAPE", "32,128,50,3").split(",")]
dev = torch.device("cuda", 0)
sc = synth.make_scene(B, level=LEVEL, tex_size=64, seed=0, device=dev)
s = 1.0 / math.tan(math.radians(15.0))
F = sc["faces"].shape[0]
first = torch.arange(B, device=dev, dtype=torch.int64) * F
num = torch.full((B,), F, device=dev, dtype=torch.int64)
sigma = 1e-4
blur = math.log(1.0 / 1e-4 - 1.0) * sigma
fuv = sc["verts_uvs"][sc["faces"]].contiguous()


def run():
    return native.render_meshes_forward(sc["verts"], sc["faces"], sc["R"], sc["T"], s, s, first, num, (H, H), blur, K,
                                        True, sc["maps"], fuv, True, sigma, (1, 1, 1), None, 1)


ref = None
for _ in range(20):
    out = run()
torch.cuda.synchronize()
img = out[0] if isinstance(out, (tuple, list)) else out
best = []
ITERS = int(os.environ.get('UR_DEV_ITERS', '200'))
for rep in range(5 if ITERS >= 200 else 1):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITERS):
        run()
    e1.record()
    torch.cuda.synchronize()
    best.append(e0.elapsed_time(e1) / ITERS * 1000)
print("UR_DEV_HYBRID=%s forward us per call (5 x 200): %s ; image checksum %.6f" % (
    os.environ.get("UR_DEV_HYBRID", "0"), [round(x, 1) for x in best], float(img.double().sum())))
