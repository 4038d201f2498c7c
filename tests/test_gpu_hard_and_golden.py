"""Two things VERDICT round 1 asked for:

* BASELINE.json configs[0] -- "64x64 hard raster K=1, fwd+bwd" -- through `Renderer(sigma=0)` (reference
  src/model/renderer.py:186-187: alpha = (dists <= 0); :46: blur_radius = ln(9999) * 0 = 0, so no barycentric
  clipping), forward AND backward, fused and two-stage, against the CPU oracle's whole path.  The gradient runs
  colour -> barycentrics -> vertices only (d(alpha)/d(dist) == 0).
* the HIP blend / UV-sampling kernels held DIRECTLY against the fixtures generated from the reference's own code
  (tests/golden/make_golden.py: `layered_rgb_blend` of renderer.py:175-202 executed from its AST node; ATen
  grid_sample as TexturesUV.sample_textures calls it), not only through the oracle.
"""
import os

import numpy as np
import pytest
import torch

from oracle import orc
from test_gpu_renderer import _setup, _oracle, rel_err, S30

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("K", [1, 8])
@pytest.mark.parametrize("N,level,img", [(1, 3, 64), (3, 3, 64), (2, 2, (40, 56))], ids=["cfg0", "batch3", "ragged"])
def test_hard_raster_forward_backward(N, level, img, K):
    H, W = (img, img) if isinstance(img, int) else img
    cot = np.random.RandomState(K).randn(N, 4, H, W).astype(np.float32)
    outs, grads = [], []
    for fused in (True, False):
        sc, rend, meshes, (verts, maps, R, T) = _setup(N, level, img, K, seed=40 + K, sigma=0.0, fused=fused)
        rs = rend.renderer.rasterizer.raster_settings
        assert rs.blur_radius == 0.0 and rs.clip_barycentric_coords is False
        out = rend(meshes, R, T)
        (out * torch.from_numpy(cot).cuda()).sum().backward()
        outs.append(out.detach())
        grads.append([t.grad.detach().cpu().numpy() for t in (verts, maps, R, T)])
    ref, (gv, gR, gT, gm) = _oracle(sc, H, W, K, S30, 0.0, cot, blur_radius=0.0)
    assert torch.equal(outs[0], outs[1])  # fused == two-stage, bit for bit
    img_np = outs[0].cpu().numpy()
    np.testing.assert_allclose(img_np, ref, atol=1e-5, rtol=0)
    alpha = img_np[:, 3]
    assert set(np.unique(alpha)) <= {0.0, 1.0} and 0.05 < alpha.mean() < 0.9  # a hard mask, and something is in it
    assert np.abs(gv).max() > 0 and np.abs(gm).max() > 0
    for g in grads:
        assert rel_err(g[1], gm) <= 1e-4
        assert rel_err(g[0], gv) <= 1e-4
        assert rel_err(g[2], gR) <= 1e-4
        assert rel_err(g[3], gT) <= 1e-4


# ---- HIP against the reference-generated fixtures, no oracle in between ---------------------------------------------
@pytest.mark.parametrize("name", ["hard", "clip", "sigmoid", "clip_s1e-2"])
def test_blend_colors_against_reference_golden(golden_dir, name):
    """ur_blend_colors / _backward vs the reference's own layered_rgb_blend outputs (renderer.py:175-202)"""
    from unicorn_amd import native
    g = np.load(os.path.join(golden_dir, f"blend_{name}.npz"))
    sigma, clip_inside = float(g["sigma"]), bool(g["clip_inside"])
    kind = native.BLEND_HARD if sigma == 0 else (native.BLEND_CLIP_INSIDE if clip_inside else native.BLEND_SIGMOID)
    t = lambda a, dt=np.float32: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).cuda()
    bg = tuple(float(v) for v in g["background"])
    col, p2f, ds = t(g["colors"]), t(g["pix_to_face"], np.int32), t(g["dists"])
    img = native.blend_colors(col, p2f, ds, None, kind, sigma, 1e-4, bg)
    np.testing.assert_allclose(img.cpu().numpy(), g["image"], atol=1e-5, rtol=0)
    # the blend's intermediates (LayeredShader debug=True) against the reference function's own debug outputs
    alpha, occ = native.blend_intermediates(p2f, ds, sigma, kind)
    np.testing.assert_allclose(alpha.cpu().numpy(), g["debug_alpha"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(occ.cpu().numpy(), g["debug_occ_alpha"], atol=1e-6, rtol=0)
    gcol, gds, _ = native.blend_colors_backward(t(g["cotangent"]), col, p2f, ds, None, kind, sigma, 1e-4, bg)
    np.testing.assert_allclose(gcol.cpu().numpy(), g["grad_colors"], atol=1e-5, rtol=1e-4)
    scale = max(1.0, float(np.abs(g["grad_dists"]).max()))
    np.testing.assert_allclose(gds.cpu().numpy(), g["grad_dists"], atol=1e-5 * scale, rtol=1e-4)


@pytest.mark.parametrize("name", ["hard", "clip", "sigmoid"])
def test_layered_rgb_blend_by_its_own_name_against_the_reference_golden(golden_dir, name):
    """unicorn_amd.renderer.layered_rgb_blend(colors, fragments, blend_params, clip_inside, debug) -- the reference
    function's name and signature (renderer.py:175-202) -- against that function's own outputs, autograd included"""
    from unicorn_amd.renderer import BlendParams, Fragments, layered_rgb_blend
    g = np.load(os.path.join(golden_dir, f"blend_{name}.npz"))
    t = lambda a, dt=np.float32: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).cuda()
    col, ds = t(g["colors"]).requires_grad_(True), t(g["dists"]).requires_grad_(True)
    frags = Fragments(t(g["pix_to_face"], np.int32), torch.zeros_like(ds), torch.zeros(*ds.shape, 3, device="cuda"), ds)
    bp = BlendParams(sigma=float(g["sigma"]), background_color=tuple(float(v) for v in g["background"]))
    img = layered_rgb_blend(col, frags, bp, clip_inside=bool(g["clip_inside"]))
    np.testing.assert_allclose(img.detach().cpu().numpy(), g["image"], atol=1e-5, rtol=0)
    img.backward(t(g["cotangent"]))
    np.testing.assert_allclose(col.grad.cpu().numpy(), g["grad_colors"], atol=1e-5, rtol=1e-4)
    scale = max(1.0, float(np.abs(g["grad_dists"]).max()))
    np.testing.assert_allclose(ds.grad.cpu().numpy(), g["grad_dists"], atol=1e-5 * scale, rtol=1e-4)
    colors_bg, alpha, occ, image = layered_rgb_blend(col.detach(), frags, bp, clip_inside=bool(g["clip_inside"]), debug=True)
    assert colors_bg.shape[-2] == col.shape[-2] + 1 and torch.equal(image, img.detach())
    np.testing.assert_allclose(alpha.cpu().numpy(), g["debug_alpha"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(occ.cpu().numpy(), g["debug_occ_alpha"], atol=1e-6, rtol=0)


def test_shade_blend_against_grid_sample_golden(golden_dir):
    """ur_shade_blend / _backward vs ATen grid_sample(flip(map), 2uv-1, bilinear, align_corners=True, 'border') as
    TexturesUV.sample_textures calls it (reached from renderer.py:162).  One opaque fragment per pixel (hard blend,
    K=1, dist < 0) turns the shader into the sampler: image rgb == texel, d image / d texel == 1."""
    from unicorn_amd import native
    g = np.load(os.path.join(golden_dir, "uvsample.npz"))
    maps, uv = g["maps"], g["uv"]  # (N,Ht,Wt,3), (N,P,2)
    N, P = uv.shape[:2]
    Ht, Wt = maps.shape[1:3]
    # face p has UV corners (1,0), (0,1), (0,0) and barycentrics (u, v, 1-u-v): sum_i b_i uv_i == (u, v)
    face_uvs = np.tile(np.array([[1, 0], [0, 1], [0, 0]], np.float32)[None], (N * P, 1, 1))
    bary = np.stack([uv[..., 0], uv[..., 1], 1 - uv[..., 0] - uv[..., 1]], -1).reshape(N, P, 1, 1, 3).astype(np.float32)
    p2f = np.arange(N * P, dtype=np.int32).reshape(N, P, 1, 1)
    ds = np.full((N, P, 1, 1), -1.0, np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    first = torch.arange(N, dtype=torch.int64, device="cuda") * P
    args = (t(p2f), t(bary), t(ds), t(maps), t(face_uvs), False, first, 0.0, (1.0, 1.0, 1.0), None, native.BLEND_HARD)
    img = native.shade_blend(*args)  # (N,4,P,1)
    np.testing.assert_allclose(img[:, :3, :, 0].permute(0, 2, 1).cpu().numpy(), g["texels"], atol=2e-6, rtol=0)
    assert torch.all(img[:, 3] == 1)
    cot = np.zeros((N, 4, P, 1), np.float32)
    cot[:, :3, :, 0] = g["cotangent"].transpose(0, 2, 1)
    gmaps, gbary, gdists = native.shade_blend_backward(t(cot), *args)
    np.testing.assert_allclose(gmaps.cpu().numpy(), g["grad_maps"], atol=1e-5, rtol=1e-5)
    gb = gbary.cpu().numpy().reshape(N, P, 3)
    np.testing.assert_allclose(gb[..., :2], g["grad_uv"], atol=1e-4, rtol=1e-4)  # face_uvs[2] = 0: d/du = gb0, d/dv = gb1
    assert (gdists == 0).all()


def test_fused_sampler_equals_the_golden_pinned_two_stage_sampler(golden_dir):
    """the FUSED kernels' sampler (k_raster<2> tail + k_render_bwd) against the same fixture: one screen-filling
    triangle per mesh whose UV corners make the interpolated uv sweep the texture; expected texels from the golden's
    maps through the golden-pinned two-stage ur_shade_blend (bit-identical images, gradients <= 1e-5)."""
    from unicorn_amd import native, ops
    g = np.load(os.path.join(golden_dir, "uvsample.npz"))
    maps = torch.from_numpy(g["maps"]).cuda().requires_grad_(True)
    N, Ht, Wt = maps.shape[:3]
    H = W = 16
    fv = torch.tensor([[[-3.0, -3.0, 2.0], [3.0, -3.0, 2.0], [0.0, 3.0, 2.0]]], device="cuda").repeat(N, 1, 1)
    fuv = torch.tensor([[[-0.1, -0.1], [1.1, 0.0], [0.5, 1.1]]], device="cuda").repeat(N, 1, 1)
    first = torch.arange(N, dtype=torch.int64, device="cuda")
    nf = torch.ones(N, dtype=torch.int64, device="cuda")
    cot = torch.randn(N, 4, H, W, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
    fused = ops.render_layered(fv, maps, first, nf, (H, W), 0.0, 1, False, fuv, False, 0.0, (1.0, 1.0, 1.0), None,
                               native.BLEND_HARD)
    fused.backward(cot)
    g_fused, maps.grad = maps.grad.clone(), None
    p2f, zb, bar, ds = ops.rasterize(fv, first, nf, (H, W), 0.0, 1, False)
    two = ops.shade_blend(p2f, bar, ds, maps, fuv, False, first, 0.0, (1.0, 1.0, 1.0), None, native.BLEND_HARD)
    two.backward(cot)
    assert torch.equal(fused.detach(), two.detach()) and torch.all(fused[:, 3] == 1)
    assert rel_err(g_fused.cpu().numpy(), maps.grad.cpu().numpy()) <= 1e-5
