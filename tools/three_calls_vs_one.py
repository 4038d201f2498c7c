"""dev measurement: the 3D step's three independent 32-image renders (unicorn.py:233 and the neighbour loop at
unicorn.py:411-413) as three Renderer calls, against ONE call on the joined batch of 96 (join_meshes_as_batch + cat of
the poses, split of the result): forward + backward, all gradients, hipGraph replay and eager.  64x64, K=50, ico3."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unicorn_amd import synth  # noqa: E402
from unicorn_amd.graphs import GraphedStep  # noqa: E402
from unicorn_amd.renderer import Renderer  # noqa: E402
from unicorn_amd.structures import Meshes, TexturesUV, join_meshes_as_batch  # noqa: E402

dev = torch.device("cuda", 0)
B, H, K = 32, 64, 50
rend = Renderer(H, cameras={"name": "fov", "fov": 30}, faces_per_pixel=K, sigma=1e-4).to(dev)
sets = []
for seed in range(3):
    sc = synth.make_scene(B, level=3, tex_size=64, seed=seed, device=dev)
    leaves = [sc[k].requires_grad_(True) for k in ("verts", "maps", "R", "T")]
    fb, ub = sc["faces"][None].expand(B, -1, -1), sc["verts_uvs"][None].expand(B, -1, -1)
    sets.append((leaves, fb, ub))
cot = [torch.randn(B, 4, H, H, device=dev) for _ in range(3)]
all_leaves = tuple(t for leaves, _, _ in sets for t in leaves)


def mesh(i):
    (v, m, _, _), fb, ub = sets[i]
    return Meshes(verts=v, faces=fb, textures=TexturesUV(m, fb, ub))


def three_calls():
    imgs = [rend(mesh(i), sets[i][0][2], sets[i][0][3]) for i in range(3)]
    torch.autograd.backward(imgs, cot)
    return imgs


def one_call():
    big = join_meshes_as_batch([mesh(i) for i in range(3)])
    R = torch.cat([sets[i][0][2] for i in range(3)])
    T = torch.cat([sets[i][0][3] for i in range(3)])
    imgs = rend(big, R, T).split(B)
    torch.autograd.backward(imgs, cot)
    return imgs


def timed(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / n)
    return sorted(ts)[2] * 1e3


def eager(fn):
    def step():
        for t in all_leaves:
            t.grad = None
        fn()
    return step



for t in all_leaves:
    t.grad = None
ia = [x.detach().clone() for x in three_calls()]
ga = [t.grad.clone() for t in all_leaves]
for t in all_leaves:
    t.grad = None
ib = [x.detach().clone() for x in one_call()]
gb = [t.grad.clone() for t in all_leaves]
same_img = all(torch.equal(x, y) for x, y in zip(ia, ib))
grad_err = max(float((x - y).abs().max() / x.abs().max().clamp_min(1e-20)) for x, y in zip(ga, gb))
out = {"images_bit_identical": same_img, "max_rel_grad_difference": grad_err,
       "eager_ms": {"three_calls": round(timed(eager(three_calls)), 4), "one_call": round(timed(eager(one_call)), 4)}}
g3 = GraphedStep(three_calls, grads_of=all_leaves)
g1 = GraphedStep(one_call, grads_of=all_leaves)
out["graph_replay_ms"] = {"three_calls": round(timed(g3), 4), "one_call": round(timed(g1), 4)}
print(json.dumps(out))
