"""ur_blend_colors / ur_blend_colors_backward through the C ABI against the CPU oracle: layered_rgb_blend over
caller-computed colours (renderer.py:172,175-202; oracle pinned by the reference's own outputs) and pytorch3d's
softmax_rgb_blend (renderer.py:205-208; oracle pinned to torch autograd of the published formula in float64 by
tests/test_oracle_golden.py::test_softmax_blend_matches_torch_autograd), then the shaders built on them."""
import numpy as np
import pytest
import torch

from oracle import orc
from unicorn_amd import synth

pytestmark = pytest.mark.gpu


def _frags(N=2, H=9, W=7, K=5, seed=0, sigma=1e-4):
    rs = np.random.RandomState(seed)
    p2f = rs.randint(-1, 40, (N, H, W, K)).astype(np.int32)
    p2f[0, 0, 0] = -1
    d = (rs.randn(N, H, W, K) * 2 * sigma).astype(np.float32)
    z = (rs.rand(N, H, W, K) * 3 + 1).astype(np.float32)
    col = rs.rand(N, H, W, K, 3).astype(np.float32)
    g = rs.randn(N, 4, H, W).astype(np.float32)
    return p2f, d, z, col, g


def rel(a, b):
    return float(np.abs(a - b).max() / max(float(np.abs(b).max()), 1e-20))


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_layered_blend_of_colours(kind):
    from unicorn_amd import native
    sigma = 0.0 if kind == 0 else 1e-4
    p2f, d, z, col, g = _frags()
    dev = "cuda:0"
    t = lambda a: torch.tensor(a, device=dev)
    img = native.blend_colors(t(col), t(p2f), t(d), None, kind, sigma, 1e-4, (0.2, 0.5, 0.9))
    ref = orc.layered_blend(col, p2f, d, sigma, (0.2, 0.5, 0.9), kind)
    np.testing.assert_allclose(img.cpu().numpy(), ref, atol=1e-5, rtol=0)
    gc, gd, gz = native.blend_colors_backward(t(g), t(col), t(p2f), t(d), None, kind, sigma, 1e-4, (0.2, 0.5, 0.9))
    rc, rd = orc.layered_blend_bwd(col, p2f, d, g, sigma, (0.2, 0.5, 0.9), kind)
    assert rel(gc.cpu().numpy(), rc) < 1e-4 and gz is None
    assert rel(gd.cpu().numpy(), rd) < 1e-4 or np.abs(rd).max() == 0


@pytest.mark.parametrize("gamma", [1e-4, 1e-2])
def test_softmax_blend(gamma):
    from unicorn_amd import native
    p2f, d, z, col, g = _frags(seed=1)
    dev = "cuda:0"
    t = lambda a: torch.tensor(a, device=dev)
    bg = (0.2, 0.5, 0.9)
    img = native.blend_colors(t(col), t(p2f), t(d), t(z), native.BLEND_SOFTMAX, 1e-4, gamma, bg, 1.0, 100.0)
    ref = orc.softmax_blend(col, p2f, d, z, 1e-4, gamma, bg)
    np.testing.assert_allclose(img.cpu().numpy(), ref, atol=1e-5, rtol=0)
    gc, gd, gz = native.blend_colors_backward(t(g), t(col), t(p2f), t(d), t(z), native.BLEND_SOFTMAX, 1e-4, gamma, bg,
                                              1.0, 100.0)
    rc, rd, rz = orc.softmax_blend_bwd(col, p2f, d, z, g, 1e-4, gamma, bg)
    assert rel(gc.cpu().numpy(), rc) < 1e-4
    assert rel(gd.cpu().numpy(), rd) < 1e-3  # fp32 exp((z_inv - z_max)/gamma) with gamma = 1e-4 is ill-conditioned
    assert rel(gz.cpu().numpy(), rz) < 1e-3


def _render_setup(N, img, K, **kw):
    from unicorn_amd.renderer import Renderer
    from unicorn_amd.structures import Meshes, TexturesUV
    dev = torch.device("cuda:0")
    sc = synth.make_scene(N, level=2, tex_size=16, seed=3)
    rend = Renderer(img, cameras={"name": "fov", "fov": 30}, faces_per_pixel=K, **kw).to(dev)
    verts = sc["verts"].to(dev).requires_grad_(True)
    maps = sc["maps"].to(dev).requires_grad_(True)
    R, T = sc["R"].to(dev).requires_grad_(True), sc["T"].to(dev).requires_grad_(True)
    fb, ub = sc["faces"].to(dev)[None].expand(N, -1, -1), sc["verts_uvs"].to(dev)[None].expand(N, -1, -1)
    return sc, rend, Meshes(verts=verts, faces=fb, textures=TexturesUV(maps, fb, ub)), (verts, maps, R, T)


def _oracle_frags(sc, H, W, K, blur):
    s = float(1.0 / np.tan(np.deg2rad(15.0)))
    N = sc["verts"].shape[0]
    ndc = orc.project_verts(sc["verts"].numpy(), sc["R"].numpy(), sc["T"].numpy(), s)
    fv = ndc[:, sc["faces"].numpy()].reshape(-1, 3, 3)
    p2f, zbuf, bary, dists = orc.rasterize(fv, N, H, W, blur, K)
    F = sc["faces"].shape[0]
    fuv = np.tile(sc["verts_uvs"].numpy()[sc["faces"].numpy()], (N, 1, 1))
    tex = orc.sample_textures(sc["maps"].numpy(), fuv, p2f, bary)
    return p2f, zbuf, bary, dists, tex


def test_soft_phong_shader_end_to_end():
    """Renderer(layered_shader=False): rasterize (HIP) -> phong shading under the default ambient light (= texels)
    -> softmax blend (HIP); forward against the oracle's stages, gradients finite and to every input."""
    sigma = 1e-4
    sc, rend, meshes, leaves = _render_setup(2, 32, 8, layered_shader=False, sigma=sigma)
    out = rend(meshes, leaves[2], leaves[3])
    assert out.shape == (2, 4, 32, 32)
    p2f, zbuf, bary, dists, tex = _oracle_frags(sc, 32, 32, 8, orc.blur_radius_for(sigma))
    ref = orc.softmax_blend(tex, p2f, dists, zbuf, sigma, 1e-4, (1.0, 1.0, 1.0))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, atol=2e-5, rtol=0)
    out.backward(torch.randn_like(out))
    for t in leaves:
        assert t.grad is not None and torch.isfinite(t.grad).all() and float(t.grad.abs().max()) > 0
    viz = rend(meshes, leaves[2], leaves[3], viz_purpose=True)  # sigma = 0 settings of the viz renderer
    assert viz.shape == (2, 4, 256, 256) and torch.isfinite(viz).all()


def test_eye_light_phong_layered():
    """the eye_light GIF configuration (renderer.py:229-234): directional light, phong shading, K=1, layered blend"""
    from unicorn_amd import lighting
    kw = dict(lights={'name': 'directional', 'direction': [[0, 0, -1]], 'ambient_color': [[0.7, 0.7, 0.7]],
                      'diffuse_color': [[0.3, 0.3, 0.3]], 'specular_color': [[0., 0., 0.]]}, shading_type='phong')
    sc, rend, meshes, leaves = _render_setup(2, 32, 1, sigma=1e-4, **kw)
    out = rend(meshes, leaves[2], leaves[3])
    p2f, zbuf, bary, dists, tex = _oracle_frags(sc, 32, 32, 1, orc.blur_radius_for(1e-4))
    # expected lit colours: the oracle's fragments through the test suite's own numpy restatement of phong shading
    # (tests/test_lighting_cpu.py, itself held to the reference's diffuse / specular by tests/golden/lighting.npz) --
    # nothing of unicorn_amd.lighting, the code under test, is on this side
    from test_lighting_cpu import _np_light
    from unicorn_amd.renderer import DirectionalLights
    from unicorn_amd.structures import Meshes
    N = 2
    cm = Meshes(verts=sc["verts"].double(), faces=sc["faces"][None].expand(N, -1, -1), textures=None)
    v, f, vn = cm.verts_packed().numpy(), cm.faces_packed().numpy(), cm.verts_normals_packed().numpy()
    mask, idx = (p2f >= 0)[..., None], np.maximum(p2f, 0)
    b64 = bary.astype(np.float64)[..., None]
    pts, nrm = (b64 * v[f][idx]).sum(-2) * mask, (b64 * vn[f][idx]).sum(-2) * mask
    R64, T64 = sc["R"].double(), sc["T"].double()
    cam = (-(T64[:, None] @ R64.transpose(1, 2))[:, 0]).numpy().reshape(N, 1, 1, 1, 3)
    amb, dif, spe = _np_light(pts, nrm, cam, DirectionalLights(**{k: v for k, v in kw['lights'].items() if k != 'name'}))
    col = (amb + dif) * tex.astype(np.float64) + spe
    ref = orc.layered_blend(col.astype(np.float32), p2f, dists, 1e-4, (1.0, 1.0, 1.0), 1)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, atol=2e-5, rtol=0)
    out.sum().backward()
    assert all(t.grad is not None and torch.isfinite(t.grad).all() for t in leaves)


def test_light_functions_against_reference_golden_on_gpu():
    """unicorn_amd.lighting.diffuse / specular on the GPU against outputs of the reference's own functions
    (pytorch3d_monkey.py:73-198; tests/golden/lighting.npz), values and gradients"""
    from test_lighting_cpu import check_light_functions_against_reference_golden
    check_light_functions_against_reference_golden("cuda", 1e-9)
