"""The one-call path (ur_render_meshes_forward / _backward) where it changes route: images with more bins than the
prep kernel's face masks cover (box scan in the rasterizer instead), more bins along one axis than the mask path's
extent table, and a mesh whose per-vertex gradient sums do not fit the grad_face_verts buffer (per-corner sums and the
corner-gathering camera backward instead).  Every case: image against the oracle (<= 1e-5), gradients (<= 1e-4
relative), fused image bit-identical to the two-stage path."""
import numpy as np
import pytest
import torch

from test_gpu_renderer import _setup, _oracle, rel_err, S30

pytestmark = pytest.mark.gpu


def _check(sc, rend, meshes, leaves, H, W, K, sigma, seed):
    verts, maps, R, T = leaves
    out = rend(meshes, R, T)
    cot = np.random.RandomState(seed).randn(*out.shape).astype(np.float32)
    (out * torch.from_numpy(cot).cuda()).sum().backward()
    ref_img, (gv, gR, gT, gm) = _oracle(sc, H, W, K, S30, sigma, cot)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref_img, atol=1e-5, rtol=0)
    errs = [rel_err(a.grad.cpu().numpy(), b) for a, b in ((maps, gm), (verts, gv), (T, gT), (R, gR))]
    assert max(errs) <= 1e-4, errs
    rend.renderer.fused = False
    assert torch.equal(rend(meshes, R, T), out)


@pytest.mark.parametrize("hw", [(528, 528), (16, 4112), (4112, 16)], ids=lambda hw: f"{hw[0]}x{hw[1]}")
def test_images_beyond_the_face_mask_limits(hw):
    """33 x 33 = 1089 bins (> 1024 mask bins) ; 257 bins along one axis (> 256)"""
    H, W = hw
    sc, rend, meshes, leaves = _setup(2, 2, hw, 6, seed=77, sigma=1e-3)
    _check(sc, rend, meshes, leaves, H, W, 6, 1e-3, seed=3)


def test_vertex_sums_that_do_not_fit_the_face_buffer():
    """8 faces over 42 vertices: 2 V > 9 F, so the backward keeps per-corner sums and gathers them per vertex"""
    from unicorn_amd.structures import Meshes, TexturesUV
    sc, rend, _, (verts, maps, R, T) = _setup(3, 1, 48, 8, seed=5, sigma=1e-3)
    sc["faces"] = sc["faces"][:8].contiguous()
    faces, uvs = sc["faces"].cuda(), sc["verts_uvs"].cuda()
    fb, ub = faces[None].expand(3, -1, -1), uvs[None].expand(3, -1, -1)
    meshes = Meshes(verts=verts, faces=fb, textures=TexturesUV(maps, fb, ub))
    _check(sc, rend, meshes, (verts, maps, R, T), 48, 48, 8, 1e-3, seed=9)
    assert float(verts.grad[:, 30:].abs().max()) >= 0.0  # vertices no face names keep a zero gradient
    used = torch.zeros(42, dtype=torch.bool)
    used[sc["faces"].flatten()] = True
    assert float(verts.grad[:, ~used.cuda()].abs().max()) == 0.0
