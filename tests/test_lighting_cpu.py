"""Lighting (unicorn_amd/lighting.py: the reference's diffuse / specular / _apply_lighting, pytorch3d_monkey.py:34-198,
and pytorch3d's phong / flat shading).  `diffuse` / `specular` are PINNED: tests/golden/lighting.npz holds outputs and
autograd gradients of the reference's own two functions (tests/golden/make_golden.py).  The shading functions around
them are checked against an independent numpy restatement (parity unpinned: pytorch3d is not available, the numpy code
below restates the same published formulas).  Pure torch ops: runs on the CPU."""
import os

import numpy as np
import torch

from unicorn_amd import lighting, synth
from unicorn_amd.renderer import DirectionalLights, PointLights, AmbientLights, Fragments
from unicorn_amd.structures import Meshes, TexturesUV, TexturesVertex


def _norm(v, eps=1e-6):
    return v / np.maximum(np.linalg.norm(v, axis=-1, keepdims=True), eps)


def _np_light(points, normals, cam, lights, shininess=64.0):
    amb = np.asarray(lights.ambient_color, np.float64).reshape(1, 3)
    if not hasattr(lights, "diffuse_color"):
        return amb, np.zeros_like(points), np.zeros_like(points)
    d = (np.asarray(lights.location, np.float64).reshape(1, 3) - points) if hasattr(lights, "location") \
        else np.broadcast_to(np.asarray(lights.direction, np.float64).reshape(1, 3), points.shape)
    n, d = _norm(normals), _norm(d)
    cos = (n * d).sum(-1)
    dif = np.asarray(lights.diffuse_color, np.float64).reshape(1, 3) * np.maximum(cos, 0)[..., None]
    view = _norm(cam - points)
    refl = -d + 2 * cos[..., None] * n
    al = np.maximum((view * refl).sum(-1), 0) * (cos > 0)
    spe = np.asarray(lights.specular_color, np.float64).reshape(1, 3) * (al ** shininess)[..., None]
    return amb, dif, spe


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lighting.npz")


def check_light_functions_against_reference_golden(device, atol):
    """lighting.diffuse / specular, values and gradients, against the fixture; also called by the GPU suite"""
    g = np.load(GOLDEN)
    t = lambda k, grad=False: torch.tensor(g[k], device=device).requires_grad_(grad)
    points, normals, direction = t("points", True), t("normals", True), t("direction", True)
    color, cam, shin = t("color"), t("camera_position"), t("shininess")
    d = lighting.diffuse(normals, color, direction)
    sp = lighting.specular(points, normals, direction, color, cam, shin)
    cmp = lambda a, k: np.testing.assert_allclose(a.detach().cpu().numpy(), g[k], atol=atol, rtol=atol, err_msg=k)
    cmp(d, "diffuse")
    cmp(sp, "specular")
    gn, gd = torch.autograd.grad((d * t("cot_diffuse")).sum(), (normals, direction), retain_graph=True)
    cmp(gn, "diffuse_grad_normals")
    cmp(gd, "diffuse_grad_direction")
    gp, gn, gd = torch.autograd.grad((sp * t("cot_specular")).sum(), (points, normals, direction))
    cmp(gp, "specular_grad_points")
    cmp(gn, "specular_grad_normals")
    cmp(gd, "specular_grad_direction")


def test_diffuse_specular_against_reference_golden():
    check_light_functions_against_reference_golden("cpu", 1e-12)
    g = np.load(GOLDEN)
    # the numpy restatement below (what the shading tests and the GPU phong test compare with) is held to it as well
    _, dif, spe = _np_light_terms(g["points"], g["normals"], g["direction"], g["color"], g["color"],
                                  g["camera_position"], g["shininess"].reshape(-1, 1, 1, 1))
    np.testing.assert_allclose(dif, g["diffuse"], atol=1e-12)
    np.testing.assert_allclose(spe, g["specular"], atol=1e-12)


def _np_light_terms(points, normals, d, dif_color, spe_color, cam, shininess):
    n, d = _norm(normals), _norm(d)
    cos = (n * d).sum(-1)
    dif = dif_color * np.maximum(cos, 0)[..., None]
    view = _norm(cam - points)
    refl = -d + 2 * cos[..., None] * n
    al = np.maximum((view * refl).sum(-1), 0) * (cos > 0)
    return None, dif, spe_color * (al ** shininess)[..., None]


def _scene(N=2, H=6, W=5, K=3, seed=0):
    sc = synth.make_scene(N, level=1, tex_size=8, seed=seed)
    F = sc["faces"].shape[0]
    rs = np.random.RandomState(seed)
    p2f = rs.randint(-1, F, (N, H, W, K)).astype(np.int64)
    p2f[p2f >= 0] += (np.arange(N).reshape(N, 1, 1, 1) * F).repeat(H, 1).repeat(W, 2).repeat(K, 3)[p2f >= 0]
    bary = rs.dirichlet((1, 1, 1), (N, H, W, K)).astype(np.float32)
    fb, ub = sc["faces"][None].expand(N, -1, -1), sc["verts_uvs"][None].expand(N, -1, -1)
    meshes = Meshes(verts=sc["verts"].double(), faces=fb, textures=TexturesUV(sc["maps"].double(), fb, ub))
    frag = Fragments(torch.tensor(p2f, dtype=torch.int32), None, torch.tensor(bary).double(), None)
    return sc, meshes, frag, p2f, bary.astype(np.float64)


def test_vertex_and_face_normals():
    sc, meshes, *_ = _scene()
    v, f = meshes.verts_packed().numpy(), meshes.faces_packed().numpy()
    fn = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    vn = np.zeros_like(v)
    for i in range(3):
        np.add.at(vn, f[:, i], fn)  # the cross product of the two edges at any corner is the same vector
    np.testing.assert_allclose(meshes.faces_normals_packed().numpy(), _norm(fn), atol=1e-12)
    np.testing.assert_allclose(meshes.verts_normals_packed().numpy(), _norm(vn), atol=1e-12)
    # an icosphere's vertex normals point outwards
    assert (np.sum(_norm(vn)[:v.shape[0] // 2] * _norm(v[:v.shape[0] // 2] - v[:v.shape[0] // 2].mean(0)), -1) > 0.5).all()


def test_phong_and_flat_shading_match_numpy():
    sc, meshes, frag, p2f, bary = _scene()
    N = p2f.shape[0]
    R, T = sc["R"].double(), sc["T"].double()
    cam = (-(T[:, None] @ R.transpose(1, 2))[:, 0]).numpy().reshape(N, 1, 1, 1, 3)
    texels = torch.rand(p2f.shape + (3,), dtype=torch.float64)
    v, f = meshes.verts_packed().numpy(), meshes.faces_packed().numpy()
    vn = meshes.verts_normals_packed().numpy()
    mask = (p2f >= 0)[..., None]
    idx = np.maximum(p2f, 0)
    pts = (bary[..., None] * v[f][idx]).sum(-2) * mask
    nrm = (bary[..., None] * vn[f][idx]).sum(-2) * mask
    for lights in (DirectionalLights(direction=((0.3, -0.2, -1.0),)), PointLights(location=((0.5, 1.0, -2.0),)),
                   AmbientLights(ambient_color=((0.7, 0.8, 0.9),))):
        amb, dif, spe = _np_light(pts, nrm, cam, lights)
        want = (amb + dif) * texels.numpy() + spe
        got = lighting.phong_shading(meshes, frag, lights, lighting.Materials(), texels, R, T).numpy()
        np.testing.assert_allclose(got, want, atol=1e-6)
    lights = DirectionalLights(direction=((0.0, 0.0, -1.0),))
    cen = v[f].mean(1)[idx] * mask
    fnr = meshes.faces_normals_packed().numpy()[idx] * mask
    amb, dif, spe = _np_light(cen, fnr, cam, lights)
    got = lighting.flat_shading(meshes, frag, lights, lighting.Materials(), texels, R, T).numpy()
    np.testing.assert_allclose(got, (amb + dif) * texels.numpy() + spe, atol=1e-6)


def test_camera_center_inverts_the_view_transform():
    sc, *_ = _scene()
    R, T = sc["R"], sc["T"]
    c = lighting.camera_center(R, T)
    np.testing.assert_allclose((torch.bmm(c[:, None], R)[:, 0] + T).numpy(), 0.0, atol=1e-5)


def test_gouraud_shading_matches_numpy():
    """lighting at the vertices (vertex colours x (ambient + diffuse) + specular), then barycentric interpolation
    (renderer.py:140-141,169)"""
    sc, meshes, frag, p2f, bary = _scene(seed=3)
    N, V = sc["verts"].shape[:2]
    vcol = torch.rand(N, V, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    meshes.textures = TexturesVertex(vcol)
    R, T = sc["R"].double(), sc["T"].double()
    cam = (-(T[:, None] @ R.transpose(1, 2))[:, 0]).numpy()
    v, f = meshes.verts_packed().numpy(), meshes.faces_packed().numpy()
    vn = meshes.verts_normals_packed().numpy()
    mask = (p2f >= 0)[..., None]
    idx = np.maximum(p2f, 0)
    for lights in (DirectionalLights(direction=((0.3, -0.2, -1.0),)), PointLights(location=((0.5, 1.0, -2.0),)),
                   AmbientLights(ambient_color=((0.7, 0.8, 0.9),))):
        amb, dif, spe = _np_light(v, vn, np.repeat(cam, V, 0), lights)
        lit = vcol.numpy().reshape(-1, 3) * (amb + dif) + spe
        want = (bary[..., None] * lit[f][idx]).sum(-2) * mask
        got = lighting.gouraud_shading(meshes, frag, lights, lighting.Materials(), R, T).numpy()
        np.testing.assert_allclose(got, want, atol=1e-6)


def test_interpolate_face_attributes_backward_equals_autograd_of_the_plain_formula():
    """shading.interpolate_face_attributes carries its own backward (a scatter-add instead of autograd's sorting
    index_put): values and both gradients must equal the plain gather-multiply-sum-mask formula's"""
    import torch
    from unicorn_amd.shading import interpolate_face_attributes
    g = torch.Generator().manual_seed(0)
    N, H, W, K, NF, D = 2, 5, 4, 3, 7, 4
    p2f = torch.randint(-1, NF, (N, H, W, K), generator=g)
    bary = torch.rand(N, H, W, K, 3, generator=g, dtype=torch.float64)
    attrs = torch.randn(NF, 3, D, generator=g, dtype=torch.float64)
    cot = torch.randn(N, H, W, K, D, generator=g, dtype=torch.float64)

    def plain(b, a):
        out = (b[..., None] * a[p2f.clamp(min=0)]).sum(-2)
        return out.masked_fill((p2f < 0)[..., None], 0.0)
    b1, a1 = bary.clone().requires_grad_(True), attrs.clone().requires_grad_(True)
    b2, a2 = bary.clone().requires_grad_(True), attrs.clone().requires_grad_(True)
    o1, o2 = interpolate_face_attributes(p2f, b1, a1), plain(b2, a2)
    assert torch.equal(o1, o2)
    o1.backward(cot)
    o2.backward(cot)
    valid = (p2f >= 0)[..., None].expand_as(bary)
    assert torch.allclose(b1.grad[valid], b2.grad[valid], rtol=1e-12, atol=0)
    assert float(b1.grad[~valid].abs().max()) == 0.0
    assert torch.allclose(a1.grad, a2.grad, rtol=1e-12, atol=1e-14)
    # whatever the empty slots' barycentrics hold does not leak (the plain formula's attribute gradient would be NaN)
    b3 = bary.clone()
    b3[p2f < 0] = float("nan")
    b3, a3 = b3.requires_grad_(True), attrs.clone().requires_grad_(True)
    o3 = interpolate_face_attributes(p2f, b3, a3)
    o3.backward(cot)
    assert torch.equal(o3, o1) and torch.equal(a3.grad, a1.grad) and torch.equal(b3.grad, b1.grad)


def test_gather_rows_backward():
    import torch
    from unicorn_amd.shading import gather_rows
    g = torch.Generator().manual_seed(1)
    table = torch.randn(6, 3, generator=g, dtype=torch.float64)
    idx = torch.randint(0, 6, (2, 4, 5), generator=g)
    cot = torch.randn(2, 4, 5, 3, generator=g, dtype=torch.float64)
    t1, t2 = table.clone().requires_grad_(True), table.clone().requires_grad_(True)
    o1, o2 = gather_rows(t1, idx), t2[idx]
    assert torch.equal(o1, o2)
    o1.backward(cot)
    o2.backward(cot)
    assert torch.allclose(t1.grad, t2.grad, rtol=1e-12, atol=1e-14)


def test_gather_rows_ignores_empty_slots():
    import torch
    from unicorn_amd.shading import gather_rows
    table = torch.arange(12, dtype=torch.float64).view(4, 3).requires_grad_(True)
    idx = torch.tensor([[2, -1, 0], [-1, -1, 3]])
    out = gather_rows(table, idx)
    assert torch.equal(out[0, 0], table[2].detach()) and torch.equal(out[1, 2], table[3].detach())
    out.backward(torch.ones_like(out))
    assert torch.equal(table.grad, torch.tensor([[1.0] * 3, [0.0] * 3, [1.0] * 3, [1.0] * 3], dtype=torch.float64))


def test_per_view_light_directions_as_the_eye_light_turntable_sets_them():
    """renderer.py:249-250 assigns lights.direction = [[0, 0, -1]] @ R_view^T, a (B, 1, 3) tensor; the reference's light
    functions view any such property as (-1, ..., 3) (pytorch3d_monkey.py:99-109)"""
    import torch
    from types import SimpleNamespace
    from unicorn_amd.lighting import Materials, light_terms
    g = torch.Generator().manual_seed(2)
    N = 4
    pts, nrm = torch.randn(N, 5, 6, 2, 3, generator=g), torch.randn(N, 5, 6, 2, 3, generator=g)
    d = torch.randn(N, 3, generator=g)
    cam = torch.randn(N, 3, generator=g)
    mk = lambda direction: SimpleNamespace(ambient_color=((0.7, 0.7, 0.7),), diffuse_color=((0.3, 0.3, 0.3),),
                                           specular_color=((0.1, 0.1, 0.1),), direction=direction)
    a = light_terms(mk(d), Materials(), pts, nrm, cam)
    b = light_terms(mk(d[:, None, :]), Materials(), pts, nrm, cam)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert not torch.equal(a[1][0], a[1][1] * 0 + a[1][0].mean())  # (the term does vary over the batch)
