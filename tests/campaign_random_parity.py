"""Randomised parity campaign (not collected by pytest: run by name, `python tests/campaign_random_parity.py [cases]`).
Random scenes -- batch, mesh level, image shape (square or not), K, sigma (soft / hard), blend mode -- through the
fused one-call path and the fragment path, against the CPU oracle: fragments bit-identical, image <= 1e-5, gradients
<= 1e-4 of their largest entry.  A wider net than the fixed cases of the suite, for changes that touch arithmetic."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import orc  # noqa: E402
from unicorn_amd import native, synth  # noqa: E402
from unicorn_amd.renderer import Renderer  # noqa: E402
from unicorn_amd.structures import Meshes, TexturesUV  # noqa: E402

S = float(1.0 / np.tan(np.deg2rad(15.0)))


def rel(a, b):
    return float(np.abs(a - b).max() / max(float(np.abs(b).max()), 1e-20))


BIG = len(sys.argv) > 3 and sys.argv[3] == "big"  # finer meshes, larger images, K beyond 64 (the long-list code)


def one_case(rng, idx):
    if BIG:
        N = int(rng.integers(1, 3))
        level = int(rng.choice([3, 4]))
        H = int(rng.choice([96, 128, 160]))
        W = H if rng.random() < 0.6 else int(rng.choice([80, 112, 144]))
        K = int(rng.choice([50, 100, 130]))
    else:
        N = int(rng.integers(1, 5))
        level = int(rng.integers(1, 4))
        H = int(rng.choice([16, 24, 32, 40, 48, 64, 72, 96]))
        W = H if rng.random() < 0.6 else int(rng.choice([16, 24, 32, 40, 56, 64, 80]))
        K = int(rng.choice([1, 3, 8, 10, 25, 50, 70]))
    sigma = float(rng.choice([0.0, 1e-5, 1e-4, 1e-4, 3e-4, 1e-3]))
    clip_inside = bool(rng.random() < 0.8) if sigma > 0 else True
    scale = float(rng.choice([0.4, 0.6, 0.9]))
    tex = int(rng.choice([8, 16, 64]))
    seed = int(rng.integers(0, 10000))
    sc = synth.make_scene(N, level=level, tex_size=tex, seed=seed, scale=scale)
    desc = f"case {idx}: N={N} level={level} {H}x{W} K={K} sigma={sigma:g} clip_inside={clip_inside} scale={scale} tex={tex} seed={seed}"
    dev = torch.device("cuda", 0)
    # fragments (MODE 0) against the oracle
    vn = orc.project_verts(sc["verts"].numpy(), sc["R"].numpy(), sc["T"].numpy(), S)
    faces = sc["faces"].numpy()
    F = faces.shape[0]
    fv = np.stack([vn[n][faces] for n in range(N)]).reshape(-1, 3, 3).astype(np.float32)
    blur = orc.blur_radius_for(sigma)
    ref = orc.rasterize(fv, N, H, W, blur, K, clip_bary=blur > 0)
    first = torch.arange(N, dtype=torch.int64, device=dev) * F
    nf = torch.full((N,), F, dtype=torch.int64, device=dev)
    out = native.rasterize_meshes(torch.from_numpy(fv).to(dev), first, nf, (H, W), blur, K, blur > 0)
    p2f, zb, bar, ds = [o.cpu().numpy() for o in out]
    assert np.array_equal(p2f, ref[0].astype(np.int32)), desc + " pix_to_face"
    assert np.array_equal(zb, ref[1]) and np.array_equal(bar, ref[2]) and np.array_equal(ds, ref[3]), desc + " fragments"
    # the fused renderer, forward + backward
    rend = Renderer((H, W), cameras={"name": "fov", "fov": 30}, faces_per_pixel=K, sigma=sigma,
                    clip_inside=clip_inside).to(dev)
    verts = sc["verts"].to(dev).requires_grad_(True)
    maps = sc["maps"].to(dev).requires_grad_(True)
    R = sc["R"].to(dev).requires_grad_(True)
    T = sc["T"].to(dev).requires_grad_(True)
    fb = sc["faces"].to(dev)[None].expand(N, -1, -1)
    ub = sc["verts_uvs"].to(dev)[None].expand(N, -1, -1)
    img = rend(Meshes(verts=verts, faces=fb, textures=TexturesUV(maps, fb, ub)), R, T)
    cot = np.random.RandomState(seed).randn(N, 4, H, W).astype(np.float32)
    img.backward(torch.from_numpy(cot).to(dev))
    uvs = np.broadcast_to(sc["verts_uvs"].numpy()[None], (N,) + tuple(sc["verts_uvs"].shape)).copy()
    args = (sc["verts"].numpy(), faces, sc["R"].numpy(), sc["T"].numpy(), sc["maps"].numpy(), uvs, H, W, K, S)
    rimg, g32 = orc.render_fwd_bwd(*args, sigma=sigma, clip_inside=clip_inside, grad_image=cot)
    err = float(np.abs(img.detach().cpu().numpy() - rimg).max())
    assert err <= 1e-5, desc + f" image {err:g}"
    g64 = None
    for i, name, got in ((0, "verts", verts.grad), (3, "maps", maps.grad), (1, "R", R.grad), (2, "T", T.grad)):  # oracle order: verts, R, T, maps
        e = rel(got.cpu().numpy(), g32[i])
        if e > 1e-4:
            # Over the bar: is the case ill-conditioned?  (A pixel outside a sliver face, barycentrics clipped onto a
            # corner: the clip-renormalisation's backward is g/s - (g.c)/s^2 = 0 in exact arithmetic, its float
            # residue is divided by the face's tiny area -- absolute noise that any float32 evaluation order
            # carries.)  Then the float32 oracle itself is that far from the float64 oracle, and the GPU is held to
            # the float64 answer within a multiple of that distance.
            if g64 is None:
                g64 = orc.render_fwd_bwd(*args, sigma=sigma, clip_inside=clip_inside, grad_image=cot, dtype=np.float64)[1]
            noise = rel(g32[i], g64[i])
            e64 = rel(got.cpu().numpy(), g64[i])
            assert e64 <= 1e-4 + 30.0 * noise, desc + f" grad {name} {e:g} (vs float64 {e64:g}, float32 oracle's own noise {noise:g})"
            print("ill-conditioned:", desc, f"grad {name}: gpu vs oracle32 {e:g}, vs oracle64 {e64:g}, oracle32 vs oracle64 {noise:g}", flush=True)
    # The same scene inside a batch big enough for the FOUR-wave build of the fused kernels (N x bins > 1024; a small
    # launch like the one above runs eight waves of 32 pixels per bin): same pictures bit for bit, same gradients.
    nbins = ((H + 15) // 16) * ((W + 15) // 16)
    reps = 1024 // (N * nbins) + 1
    if reps > 1 and reps * N <= 400:
        g8 = [t.grad.clone() for t in (verts, maps, R, T)]
        big = [t.detach().repeat(reps, *([1] * (t.dim() - 1))).requires_grad_(True) for t in (verts, maps, R, T)]
        fb2 = sc["faces"].to(dev)[None].expand(reps * N, -1, -1)
        ub2 = sc["verts_uvs"].to(dev)[None].expand(reps * N, -1, -1)
        img2 = rend(Meshes(verts=big[0], faces=fb2, textures=TexturesUV(big[1], fb2, ub2)), big[2], big[3])
        assert torch.equal(img2[:N], img) and torch.equal(img2[-N:], img), desc + " four-wave image != eight-wave image"
        cot2 = torch.zeros_like(img2)
        cot2[:N] = torch.from_numpy(cot).to(dev)
        img2.backward(cot2)
        for name, a, b in zip(("verts", "maps", "R", "T"), g8, big):
            e = rel(b.grad[:N].cpu().numpy(), a.cpu().numpy())
            assert e <= 5e-5, desc + f" four-wave vs eight-wave grad {name} {e:g}"
            assert float(b.grad[N:].abs().max()) == 0.0, desc + " gradient leaked into a mesh with a zero cotangent"
    # Pose tiling: the first mesh under all N poses, as an index (repeat -> n_meshes = 1) and as N
    # physical copies: same pictures bit for bit, the one mesh's gradients summed over its poses.
    if N > 1:
        from unicorn_amd.structures import repeat
        leaves = [t.detach()[:1].clone().requires_grad_(True) for t in (verts, maps)]
        Rt, Tt = R.detach().clone().requires_grad_(True), T.detach().clone().requires_grad_(True)
        one = Meshes(verts=leaves[0], faces=fb[:1], textures=TexturesUV(leaves[1], fb[:1], ub[:1]))
        tiled = rend(repeat(one, N), Rt, Tt)
        tiled.backward(torch.from_numpy(cot).to(dev))
        got = [leaves[0].grad.clone(), leaves[1].grad.clone(), Rt.grad.clone(), Tt.grad.clone()]
        for t in leaves + [Rt, Tt]:
            t.grad = None
        copies = Meshes(verts=leaves[0].repeat(N, 1, 1), faces=fb, textures=TexturesUV(leaves[1].repeat(N, 1, 1, 1), fb, ub))
        img_c = rend(copies, Rt, Tt)
        assert torch.equal(img_c, tiled), desc + " tiled image != copied image"
        img_c.backward(torch.from_numpy(cot).to(dev))
        for name, a, b in zip(("verts", "maps", "R", "T"), got, (leaves[0].grad, leaves[1].grad, Rt.grad, Tt.grad)):
            e = rel(a.cpu().numpy(), b.cpu().numpy())
            assert e <= 5e-5, desc + f" tiled vs copied grad {name} {e:g}"
    return desc


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    for i in range(n):
        d = one_case(rng, i)
        if i % 10 == 0:
            print("ok", d, flush=True)
    print(f"{n} random cases agree with the oracle")


if __name__ == "__main__":
    main()
