"""A UNICORN-shaped data-parallel training step around the renderer (BASELINE.json configs[2]: encoder + renderer,
64x64, DDP over the GPUs of one node).

What the reference does per iteration (unicorn.py:217-244, trainer.py:142-158), reduced to its skeleton:
image -> encoder -> {vertex displacements of the template icosphere, UV texture, pose} -> Renderer -> compose over the
background -> reconstruction loss -> backward -> optimizer step, with the alternate optimisation of unicorn.py:226-230
(pose step: mesh detached, grads to the pose head only; 3D step: pose detached) and one process per GPU under
DistributedDataParallel (trainer.py:260-264).  The CNN here is a small conv stack standing in for the ResNet trunk
(torchvision is not in this image); it stays on PyTorch-ROCm, as north_star says.  The renderer has no parameters
and needs no collective: the only communication is DDP's gradient all-reduce (RCCL over xGMI with backend "nccl").
Half the parameters get no gradient in every step, so DDP needs find_unused_parameters=True.

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_step.py --steps 20
"""
import argparse
import math
import os
import sys
import time

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from unicorn_amd import dist as udist  # noqa: E402
from unicorn_amd import synth  # noqa: E402
from unicorn_amd.renderer import Renderer  # noqa: E402
from unicorn_amd.structures import Meshes, TexturesUV  # noqa: E402


class TinyUnicorn(nn.Module):
    def __init__(self, img_size=64, level=3, faces_per_pixel=25, tex_size=64, latent=128, scale=0.6):
        super().__init__()
        chans = [3, 32, 64, 128, latent]
        self.encoder = nn.Sequential(*[m for i in range(4) for m in
                                       (nn.Conv2d(chans[i], chans[i + 1], 3, stride=2, padding=1), nn.ReLU())],
                                     nn.AdaptiveAvgPool2d(1), nn.Flatten())
        verts, faces = synth.icosphere(level)
        verts = verts * (torch.tensor(synth.SCALE_ELLIPSE) * scale)                     # unicorn.py:86-92
        self.register_buffer("template", verts)
        self.register_buffer("faces", faces)
        self.register_buffer("uvs", synth.spherical_uvs(verts))                         # unicorn.py:93
        self.tex_size = tex_size
        self.deform = nn.Linear(latent, verts.shape[0] * 3)
        self.texture = nn.Linear(latent, tex_size * tex_size * 3)
        self.pose = nn.Linear(latent, 6)
        for head, gain in ((self.deform, 1e-2), (self.pose, 1e-1)):
            nn.init.normal_(head.weight, std=gain / math.sqrt(latent))
            nn.init.zeros_(head.bias)
        self.renderer = Renderer(img_size, cameras={"name": "fov", "fov": 30}, faces_per_pixel=faces_per_pixel,
                                 sigma=1e-4)
        self.pose_step = True

    def _apply(self, fn, *a, **kw):   # the renderer holds non-module state (cameras, lights): unicorn.py:211-215
        super()._apply(fn, *a, **kw)
        self.renderer.to(self.template.device)
        return self

    def predict(self, imgs):
        z = self.encoder(imgs)
        B = z.shape[0]
        verts = self.template[None] + 0.1 * torch.tanh(self.deform(z)).view(B, -1, 3)
        maps = torch.sigmoid(self.texture(z)).view(B, self.tex_size, self.tex_size, 3)
        p = self.pose(z)
        azim, elev, roll = 180.0 * torch.tanh(p[:, 0]), 10.0 + 20.0 * torch.tanh(p[:, 1]), 30.0 * torch.tanh(p[:, 2])
        R = synth.pose_to_R(azim, elev, roll)                                            # unicorn.py:306
        T = torch.tensor(synth.PRIOR_TRANSLATION, device=z.device)[None] + 0.3 * torch.tanh(p[:, 3:])
        return verts, maps, R, T

    def forward(self, imgs):
        verts, maps, R, T = self.predict(imgs)
        if self.pose_step:                                                               # unicorn.py:226-230
            verts, maps = verts.detach(), maps.detach()
        else:
            R, T = R.detach(), T.detach()
        B = imgs.shape[0]
        fb, ub = self.faces[None].expand(B, -1, -1), self.uvs[None].expand(B, -1, -1)
        rec = self.renderer(Meshes(verts=verts, faces=fb, textures=TexturesUV(maps, fb, ub)), R, T)   # unicorn.py:233
        fg, alpha = rec[:, :3], rec[:, 3:]
        out = fg * alpha + (1.0 - alpha) * 1.0                                           # unicorn.py:234, white bkg
        return ((out - imgs) ** 2).flatten(1).mean(1)                                    # per-image loss

    def iter_step(self):                                                                 # unicorn.py:450-453
        self.pose_step = not self.pose_step


def synthetic_targets(n, img_size, seed, device):
    """n target images: renders of random synthetic scenes (no datasets offline)"""
    sc = synth.make_scene(n, level=3, tex_size=32, seed=seed, device=device)
    rend = Renderer(img_size, cameras={"name": "fov", "fov": 30}, faces_per_pixel=10, sigma=1e-4).to(device)
    fb, ub = sc["faces"][None].expand(n, -1, -1), sc["verts_uvs"][None].expand(n, -1, -1)
    with torch.no_grad():
        return rend(Meshes(sc["verts"], fb, TexturesUV(sc["maps"], fb, ub)), sc["R"], sc["T"])[:, :3].contiguous()


def train_steps(model, imgs, steps, lr=0.05):
    """`steps` iterations of forward / loss.mean().backward() / SGD step / iter_step (trainer.py:142-158);
    `model` is the bare module or its DDP wrapper.  Returns the per-step mean losses."""
    core = model.module if hasattr(model, "module") else model
    opt = torch.optim.SGD(core.parameters(), lr=lr)
    losses = []
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        loss = model(imgs).mean()        # DDP averages the ranks' gradients: the global-batch mean (trainer.py:148-151)
        loss.backward()
        opt.step()
        core.iter_step()
        losses.append(float(loss.detach()))
    return losses


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32, help="per GPU (car.yml: 32)")
    ap.add_argument("--img", type=int, default=64)
    a = ap.parse_args()
    rank, local, world = udist.env_world()
    if world > 1:
        udist.isolate_miopen_cache(rank)                  # before the first convolution: no shared kernel-cache locks
    # rehearsal on a box with fewer GPUs than ranks (UNICORN_BENCH_BACKEND=gloo, as in bench.py): the ranks share the
    # GPUs and meet over gloo; on a node every rank has its own GPU and DDP's all-reduce runs on RCCL ("nccl")
    backend = os.environ.get("UNICORN_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    udist.init(backend)
    dev = torch.device("cuda", local)
    torch.manual_seed(0)                                  # same initial weights on every rank
    model = TinyUnicorn(a.img).to(dev)
    if world > 1:
        model = nn.parallel.DistributedDataParallel(model, device_ids=[local], find_unused_parameters=True)
    imgs = synthetic_targets(a.batch, a.img, seed=udist.shard_seed(7, rank), device=dev)
    train_steps(model, imgs, 2)                           # warm-up
    udist.fence(world, cuda=True)
    t0 = time.perf_counter()
    losses = train_steps(model, imgs, a.steps)
    udist.fence(world, cuda=True)
    dt = udist.max_over_ranks(time.perf_counter() - t0, world, cuda=True)
    if rank == 0:
        shared = world > torch.cuda.device_count()
        print(f"{world} rank(s){' SHARING ' + str(torch.cuda.device_count()) + ' GPU(s) over ' + backend if shared else ''}: "
              f"{world * a.batch * a.steps / dt:.0f} images/s through the whole training step "
              f"(encoder + renderer + DDP all-reduce), loss {losses[0]:.4f} -> {losses[-1]:.4f}")
    udist.finish(world)


if __name__ == "__main__":
    main()
