"""`TexturesVertex` rendering (reference: src/model/renderer.py:273 voxel GIFs, src/model/unicorn.py:488-494 coloured
template meshes; pytorch3d TexturesVertex.sample_textures = barycentric interpolation of the three corner colours) and
gouraud shading (renderer.py:140-141,169) through `Renderer`, against the CPU oracle's rasterizer + layered blend with
the colour stage restated in numpy; and the cameras' screen transform against the rasterizer's own pixel grid."""
import numpy as np
import pytest
import torch

from oracle import orc
from unicorn_amd import synth

pytestmark = pytest.mark.gpu

S30 = float(1.0 / np.tan(np.deg2rad(15.0)))


def rel_err(a, b):
    return float(np.abs(a - b).max() / max(float(np.abs(b).max()), 1e-20))


def _oracle_vertex_colour_render(sc, vcol, H, W, K, sigma, cot):
    """project -> rasterize -> colours = sum_i bary_i * vcol[face corner i] -> layered blend, and the backward of all
    of it: (image, grad_vcol, grad_verts)"""
    verts, faces, R, T = sc["verts"].numpy(), sc["faces"].numpy(), sc["R"].numpy(), sc["T"].numpy()
    N, V = verts.shape[:2]
    F = faces.shape[0]
    blur = orc.blur_radius_for(sigma)
    vn = orc.project_verts(verts, R, T, S30)
    fv = np.stack([vn[n][faces] for n in range(N)]).reshape(-1, 3, 3)
    p2f, zb, bar, ds = orc.rasterize(fv, N, H, W, blur, K)
    attrs = np.stack([vcol[n][faces] for n in range(N)]).reshape(-1, 3, 3)  # (NF, corner, rgb)
    mask = (p2f >= 0)[..., None]
    idx = np.maximum(p2f, 0)
    col = ((bar[..., None] * attrs[idx]).sum(-2) * mask).astype(np.float32)
    mode = orc.blend_mode(sigma, True)
    img = orc.layered_blend(col, p2f, ds, sigma, (1.0, 1.0, 1.0), mode)
    gcol, gds = orc.layered_blend_bwd(col, p2f, ds, cot, sigma, (1.0, 1.0, 1.0), mode)
    gcol = gcol * mask
    gbar = (gcol[..., None, :] * attrs[idx]).sum(-1).astype(np.float32)          # (N,H,W,K,3 corners)
    gattr = np.zeros_like(attrs)
    np.add.at(gattr, idx[mask[..., 0]], (bar[..., None] * gcol[..., None, :])[mask[..., 0]])
    gvcol = np.zeros_like(vcol)
    for n in range(N):
        np.add.at(gvcol[n], faces, gattr.reshape(N, F, 3, 3)[n])
    gfv = orc.rasterize_backward(fv, p2f, np.zeros_like(zb), gbar, gds, blur > 0)
    g_ndc = np.zeros_like(vn)
    for n in range(N):
        np.add.at(g_ndc[n], faces, gfv.reshape(N, F, 3, 3)[n])
    gv, _, _ = orc.project_verts_bwd(verts, R, T, S30, g_ndc)
    return img, gvcol, gv


@pytest.mark.parametrize("sigma,K", [(1e-4, 10), (0.0, 1)], ids=["soft", "hard"])
def test_textures_vertex_render_matches_oracle(sigma, K):
    from unicorn_amd.renderer import Renderer
    from unicorn_amd.structures import Meshes, TexturesVertex
    N, H, W = 2, 32, 40
    dev = torch.device("cuda:0")
    sc = synth.make_scene(N, level=2, seed=17)
    V = sc["verts"].shape[1]
    vcol_np = np.random.RandomState(3).rand(N, V, 3).astype(np.float32)
    rend = Renderer((H, W), cameras={"name": "fov", "fov": 30}, faces_per_pixel=K, sigma=sigma).to(dev)
    verts = sc["verts"].to(dev).requires_grad_(True)
    vcol = torch.from_numpy(vcol_np).to(dev).requires_grad_(True)
    meshes = Meshes(verts=verts, faces=sc["faces"].to(dev)[None].expand(N, -1, -1), textures=TexturesVertex(vcol))
    out = rend(meshes, sc["R"].to(dev), sc["T"].to(dev))
    assert out.shape == (N, 4, H, W)
    cot = np.random.RandomState(4).randn(N, 4, H, W).astype(np.float32)
    (out * torch.from_numpy(cot).to(dev)).sum().backward()
    ref, gvcol, gv = _oracle_vertex_colour_render(sc, vcol_np, H, W, K, sigma, cot)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, atol=1e-5, rtol=0)
    assert 0.05 < ref[:, 3].mean() < 0.9
    assert rel_err(vcol.grad.cpu().numpy(), gvcol) <= 1e-4
    assert rel_err(verts.grad.cpu().numpy(), gv) <= 1e-4


def test_textures_vertex_list_constructor_and_viz_path():
    """utils/mesh.py:165 builds TexturesVertex from a list; renderer.py:273 renders such a mesh through the viz path"""
    from unicorn_amd.renderer import Renderer
    from unicorn_amd.structures import Meshes, TexturesVertex
    dev = torch.device("cuda:0")
    sc = synth.make_scene(1, level=2, seed=2)
    V = sc["verts"].shape[1]
    tex = TexturesVertex(verts_features=[torch.ones(V, 3, device=dev) * 0.8])
    assert tex.verts_features_list()[0].shape == (V, 3)
    mesh = Meshes(sc["verts"].to(dev), sc["faces"].to(dev)[None], tex)
    rend = Renderer(32, cameras={"name": "fov", "fov": 30}, viz_size=(24, 24)).to(dev)
    img = rend(mesh.extend(3), sc["R"].to(dev).expand(3, -1, -1), sc["T"].to(dev).expand(3, -1), viz_purpose=True)
    assert img.shape == (3, 4, 24, 24)
    inside = img[:, 3] == 1
    assert inside.any() and torch.allclose(img[:, :3].permute(0, 2, 3, 1)[inside], torch.tensor(0.8, device=dev))
    assert torch.equal(img[0], img[1])


def test_gouraud_shading_end_to_end():
    """shading_type='gouraud' (renderer.py:140-141,169): lighting at the vertices, interpolated, then layered blend"""
    from unicorn_amd.renderer import Renderer
    from unicorn_amd.structures import Meshes, TexturesVertex
    N, H, K, sigma = 2, 32, 6, 1e-4
    dev = torch.device("cuda:0")
    sc = synth.make_scene(N, level=2, seed=23)
    V = sc["verts"].shape[1]
    vcol_np = np.random.RandomState(5).rand(N, V, 3).astype(np.float32)
    lights = {"name": "directional", "direction": [[0.3, -0.2, -1.0]], "ambient_color": [[0.6, 0.6, 0.6]],
              "diffuse_color": [[0.4, 0.3, 0.2]], "specular_color": [[0.2, 0.2, 0.2]]}
    rend = Renderer(H, cameras={"name": "fov", "fov": 30}, lights=lights, faces_per_pixel=K, sigma=sigma,
                    shading_type="gouraud").to(dev)
    meshes = Meshes(sc["verts"].to(dev), sc["faces"].to(dev)[None].expand(N, -1, -1),
                    TexturesVertex(torch.from_numpy(vcol_np).to(dev)))
    R, T = sc["R"].to(dev), sc["T"].to(dev)
    out = rend(meshes, R, T)
    # numpy restatement of the per-vertex lighting (pytorch3d_monkey.py:34-198 with Materials() = ones, shininess 64)
    v = sc["verts"].numpy().astype(np.float64)
    f = sc["faces"].numpy()
    lit = np.zeros((N, V, 3))
    for n in range(N):
        fn = np.cross(v[n][f[:, 1]] - v[n][f[:, 0]], v[n][f[:, 2]] - v[n][f[:, 0]])
        vn = np.zeros((V, 3))
        for i in range(3):
            np.add.at(vn, f[:, i], fn)
        nrm = vn / np.maximum(np.linalg.norm(vn, axis=-1, keepdims=True), 1e-6)
        d = np.array(lights["direction"][0])
        d = d / np.linalg.norm(d)
        cos = nrm @ d
        dif = np.array(lights["diffuse_color"][0]) * np.maximum(cos, 0)[:, None]
        cam = -(sc["T"][n].numpy().astype(np.float64) @ sc["R"][n].numpy().astype(np.float64).T)
        view = cam - v[n]
        view = view / np.maximum(np.linalg.norm(view, axis=-1, keepdims=True), 1e-6)
        refl = -d + 2 * cos[:, None] * nrm
        al = np.maximum((view * refl).sum(-1), 0) * (cos > 0)
        spe = np.array(lights["specular_color"][0]) * (al ** 64)[:, None]
        lit[n] = vcol_np[n] * (np.array(lights["ambient_color"][0]) + dif) + spe
    ref, _, _ = _oracle_vertex_colour_render(sc, lit.astype(np.float32), H, H, K, sigma,
                                             np.zeros((N, 4, H, H), np.float32))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, atol=2e-5, rtol=0)


@pytest.mark.parametrize("H,W", [(64, 64), (48, 80), (80, 48)])
def test_transform_points_screen_is_the_rasterizers_pixel_grid(H, W):
    """kp_eval.py:91-104 projects vertices with cameras.transform_points_screen and compares with pixel-space keypoints:
    its (x, y) must be the column / row the rasterizer puts the same 3D point in.  A tiny triangle around a point is
    hard-rasterized: the one pixel it covers must be round(transform_points_screen(point))."""
    from unicorn_amd.renderer import Renderer
    from unicorn_amd.structures import Meshes, TexturesVertex
    dev = torch.device("cuda:0")
    rend = Renderer((H, W), cameras={"name": "fov", "fov": 30}).to(dev)
    rs = np.random.RandomState(H + W)
    N = 6
    pix = np.stack([rs.randint(2, W - 2, N), rs.randint(2, H - 2, N)], -1)  # (col, row) targets
    sc = synth.make_scene(N, level=1, seed=1)
    R, T = sc["R"].to(dev), sc["T"].to(dev)
    # un-project the pixel centres at view depth z: x_ndc = 1 - (2 col + 1)/W scaled for non-square images
    z = torch.full((N,), 2.5)
    sx = max(W / H, 1.0)
    sy = max(H / W, 1.0)
    x_ndc = torch.tensor(sx * (1 - (2 * pix[:, 0] + 1) / W), dtype=torch.float32)
    y_ndc = torch.tensor(sy * (1 - (2 * pix[:, 1] + 1) / H), dtype=torch.float32)
    view = torch.stack([x_ndc * z / S30, y_ndc * z / S30, z], -1).to(dev)
    world = torch.bmm((view - T)[:, None], R.transpose(1, 2))  # (N,1,3): X = (view - T) R^T
    scr = rend.cameras.transform_points_screen(world, image_size=(H, W), R=R, T=T)
    np.testing.assert_allclose(scr[:, 0, :2].cpu().numpy(), pix.astype(np.float32), atol=2e-3)
    np.testing.assert_allclose(scr[:, 0, 2].cpu().numpy(), 2.5, atol=1e-4)
    # rasterizer: a triangle of half a pixel around the point covers that pixel centre and no other
    e = 0.6 * z[0] / S30 * (2.0 / min(H, W)) * 0.5
    offs = torch.tensor([[-e, -e, 0.0], [e, -e, 0.0], [0.0, e, 0.0]], device=dev)
    tri_view = view[:, None] + offs[None]
    tri_world = torch.bmm(tri_view - T[:, None], R.transpose(1, 2))
    faces = torch.tensor([[0, 1, 2]], device=dev)[None].expand(N, -1, -1)
    meshes = Meshes(tri_world, faces, TexturesVertex(torch.ones(N, 3, 3, device=dev)))
    from unicorn_amd.renderer import RasterizationSettings
    frag = rend.renderer.rasterizer(meshes, R=R, T=T, raster_settings=RasterizationSettings(
        image_size=(H, W), blur_radius=0.0, faces_per_pixel=1, cull_backfaces=False))
    p2f = frag.pix_to_face[..., 0].cpu().numpy()
    for n in range(N):
        rows, cols = np.nonzero(p2f[n] >= 0)
        assert list(zip(cols, rows)) == [(pix[n, 0], pix[n, 1])], (n, list(zip(cols, rows)), pix[n])


def test_batch_of_unequal_meshes_with_vertex_colours():
    """meshes of different sizes in one batch (lists; pytorch3d's TexturesVertex takes a list of (V_n, C) tensors): each
    image, and each mesh's gradients, must be what rendering that mesh alone gives (images are independent units)"""
    from unicorn_amd.renderer import Renderer
    from unicorn_amd.structures import Meshes, TexturesVertex
    dev = torch.device("cuda:0")
    H = W = 48
    sc = synth.make_scene(3, level=3, seed=21)
    v2, f2 = synth.icosphere(2)
    verts = [sc["verts"][0].to(dev), (v2 * 0.45).to(dev), sc["verts"][2].to(dev)]
    faces = [sc["faces"].to(dev), f2.to(dev), sc["faces"].to(dev)]
    rs = np.random.RandomState(5)
    cols = [torch.from_numpy(rs.rand(v.shape[0], 3).astype(np.float32)).to(dev) for v in verts]
    R, T = sc["R"].to(dev), sc["T"].to(dev)
    cot = torch.from_numpy(rs.randn(3, 4, H, W).astype(np.float32)).to(dev)
    rend = Renderer((H, W), cameras={"name": "fov", "fov": 30}, faces_per_pixel=10, sigma=1e-4).to(dev)

    def leaves():
        return [v.clone().requires_grad_(True) for v in verts], [c.clone().requires_grad_(True) for c in cols]
    vb, cb = leaves()
    out = rend(Meshes(vb, faces, TexturesVertex(cb)), R, T)
    assert out.shape == (3, 4, H, W)
    out.backward(cot)
    for i in range(3):
        vi, ci = leaves()
        alone = rend(Meshes([vi[i]], [faces[i]], TexturesVertex([ci[i]])), R[i:i + 1], T[i:i + 1])
        # (a list batch is projected by torch ops in the kernel's operation order, a single mesh by k_project_faces)
        assert torch.equal(alone[0], out[i].detach()), f"image {i}: {float((alone[0] - out[i].detach()).abs().max()):g}"
        alone.backward(cot[i:i + 1])
        assert rel_err(vb[i].grad.cpu().numpy(), vi[i].grad.cpu().numpy()) <= 1e-4
        assert rel_err(cb[i].grad.cpu().numpy(), ci[i].grad.cpu().numpy()) <= 1e-4
        assert 0.02 < float(out[i, 3].detach().mean()) < 0.9
