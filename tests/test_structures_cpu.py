"""The `Meshes` / texture / camera surface the reference's callers use, on the CPU: plain tensor
bookkeeping, no kernels.  Call sites: unicorn.py:92-93 (scale_verts, get_mesh_verts_faces), :262-266 (extend,
offset_verts_, scale_verts_), :389,407 (`meshes[i]`-style indexing and detach in the neighbour swap), :499
(offset_verts), :512-513 (extend); utils/mesh.py:21-71 (normalize, repeat); renderer.py:247 (look_at_view_transform);
kp_eval.py:91-104 (transform_points_screen)."""
import math

import numpy as np
import pytest
import torch

from unicorn_amd import synth
from unicorn_amd.cameras import FoVPerspectiveCameras, PerspectiveCameras, look_at_view_transform
from unicorn_amd.structures import Meshes, TexturesUV, TexturesVertex, join_meshes_as_batch, repeat


def _template(level=1):
    v, f = synth.icosphere(level)
    return Meshes(v[None], f[None])


def test_predict_meshes_call_sequence():
    """unicorn.py:262-266: template.extend(B); offset_verts_(packed displacements); textures set; scale_verts_((B,3))"""
    tmpl = _template().scale_verts(torch.tensor([[1.0, 0.7, 0.7]]) * 0.6)  # unicorn.py:88-92: (1,3) scale
    verts, faces = tmpl.get_mesh_verts_faces(0)
    V, B = verts.shape[0], 4
    assert torch.allclose(verts, synth.icosphere(1)[0] * torch.tensor([0.6, 0.42, 0.42]))
    g = torch.Generator().manual_seed(0)
    disp = torch.randn(B * V, 3, generator=g, requires_grad=True)
    scales = (1 + 0.1 * torch.randn(B, 3, generator=g)).requires_grad_(True)
    meshes = tmpl.extend(B)
    assert len(meshes) == B and meshes.verts_padded().shape == (B, V, 3)
    ret = meshes.offset_verts_(disp)
    assert ret is meshes  # in place, like pytorch3d
    maps = torch.rand(B, 8, 8, 3, generator=g)
    uvs = synth.spherical_uvs(verts)[None]
    meshes.textures = TexturesUV(maps, faces[None].expand(B, -1, -1), uvs.expand(B, -1, -1))
    meshes.scale_verts_(scales)
    want = (verts[None] + disp.view(B, V, 3)) * scales[:, None]
    assert torch.allclose(meshes.verts_padded(), want)
    assert torch.equal(meshes.verts_packed(), meshes.verts_padded().reshape(-1, 3))
    meshes.verts_padded().sum().backward()  # autograd reaches both predictions
    assert disp.grad is not None and scales.grad is not None and scales.grad.shape == (B, 3)
    # faces_packed offsets every mesh's indices by its first vertex
    fp = meshes.faces_packed()
    assert fp.shape == (B * faces.shape[0], 3) and int(fp.max()) == B * V - 1
    assert torch.equal(fp[faces.shape[0]:2 * faces.shape[0]], faces + V)
    assert torch.equal(meshes.num_verts_per_mesh(), torch.full((B,), V))
    assert torch.equal(meshes.verts_packed_to_mesh_idx(), torch.arange(B).repeat_interleave(V))


def test_offset_verts_returns_a_new_mesh_and_accepts_a_vector():
    tmpl = _template()
    V = tmpl.verts_packed().shape[0]
    moved = tmpl.offset_verts(torch.ones(V, 3))  # unicorn.py:499
    assert moved is not tmpl and torch.allclose(moved.verts_packed(), tmpl.verts_packed() + 1)
    assert torch.allclose(tmpl.offset_verts(torch.tensor([1.0, 2.0, 3.0])).verts_packed(),
                          tmpl.verts_packed() + torch.tensor([1.0, 2.0, 3.0]))
    assert torch.allclose(tmpl.scale_verts(2.0).verts_packed(), 2 * tmpl.verts_packed())


def test_extend_is_repeat_interleave_and_repeat_is_tile():
    """pytorch3d extend: (m0, m0, m1, m1); the reference's repeat (utils/mesh.py:43-71): (m0, m1, m0, m1)"""
    v, f = synth.icosphere(1)
    verts = torch.stack([v, 2 * v, 3 * v])
    tex = TexturesVertex(torch.stack([torch.full_like(v, c) for c in (0.1, 0.2, 0.3)]))
    m = Meshes(verts, f[None].expand(3, -1, -1), tex)
    e, r = m.extend(2), repeat(m, 2)
    scale = lambda mm: [round(float(x.abs().max() / v.abs().max())) for x in mm.verts_list()]
    assert scale(e) == [1, 1, 2, 2, 3, 3] and scale(r) == [1, 2, 3, 1, 2, 3]
    assert [round(float(t[0, 0]), 1) for t in e.textures.verts_features_padded()] == [0.1, 0.1, 0.2, 0.2, 0.3, 0.3]
    assert [round(float(t[0, 0]), 1) for t in r.textures.verts_features_padded()] == [0.1, 0.2, 0.3, 0.1, 0.2, 0.3]
    assert repeat(m, 1) is m


def test_indexing_detach_and_list_built_meshes():
    v, f = synth.icosphere(1)
    verts = [v.clone().requires_grad_(True), (2 * v).requires_grad_(True), (3 * v).requires_grad_(True)]
    maps = torch.rand(3, 4, 4, 3, requires_grad=True)
    uvs = synth.spherical_uvs(v)
    m = Meshes(verts, [f.clone() for _ in range(3)], TexturesUV(maps, f[None].repeat(3, 1, 1), uvs[None].repeat(3, 1, 1)))
    assert m.equisized and len(m) == 3 and not m.isempty()
    one = m[1]
    assert len(one) == 1 and torch.equal(one.verts_padded()[0], verts[1]) and one.textures.maps_padded().shape[0] == 1
    sub = m[[2, 0]]
    assert torch.equal(sub.verts_list()[0], verts[2]) and torch.equal(sub.textures.maps_padded()[1], maps[0])
    assert torch.equal(m[torch.tensor([True, False, True])].verts_list()[1], verts[2])
    assert len(m[1:]) == 2
    d = m.detach()
    assert not d.verts_padded().requires_grad and not d.textures.maps_padded().requires_grad
    assert m.verts_padded().requires_grad and torch.equal(d.verts_padded(), m.verts_padded())
    vv, ff = m.get_mesh_verts_faces(2)
    assert vv is verts[2] and torch.equal(ff, f)
    j = join_meshes_as_batch([m[0], m[2]])
    assert len(j) == 2 and torch.equal(j.verts_list()[1], verts[2])
    # heterogeneous batch: not equisized, packed views still consistent
    v2, f2 = synth.icosphere(2)
    h = Meshes([v, v2], [f, f2])
    assert not h.equisized and h.verts_packed().shape[0] == v.shape[0] + v2.shape[0]
    assert torch.equal(h.faces_packed()[f.shape[0]:], f2 + v.shape[0])
    assert torch.equal(h.mesh_to_faces_packed_first_idx(), torch.tensor([0, f.shape[0]]))


def test_edges_of_a_closed_mesh():
    """edges_packed / faces_packed_to_edges_packed feed normal_consistency (utils/mesh.py:92-139)"""
    m = _template(2)
    V, F = m.verts_packed().shape[0], m.faces_packed().shape[0]
    e, f2e = m.edges_packed(), m.faces_packed_to_edges_packed()
    assert e.shape == (3 * F // 2, 2) and V - e.shape[0] + F == 2  # Euler
    assert (e[:, 0] < e[:, 1]).all() and f2e.shape == (F, 3)
    faces = m.faces_packed()
    for i, (a, b) in enumerate([(1, 2), (2, 0), (0, 1)]):  # edge i is opposite to corner i
        got = e[f2e[:, i]]
        want = torch.stack([faces[:, a], faces[:, b]], 1).sort(1).values
        assert torch.equal(got, want)
    assert torch.bincount(f2e.flatten()).eq(2).all()  # every edge has exactly two faces


def test_look_at_view_transform():
    """renderer.py:247: turntable cameras.  R is orthonormal with det +1, the camera sits at distance `dist` from the
    origin, looks at it (the origin projects to view (0, 0, dist)) and keeps +Y up."""
    azim = torch.linspace(-180, 180, 9)
    R, T = look_at_view_transform(dist=2.5, elev=30.0, azim=azim)
    assert R.shape == (9, 3, 3) and T.shape == (9, 3)
    eye = torch.eye(3).expand(9, 3, 3)
    assert torch.allclose(R @ R.transpose(1, 2), eye, atol=1e-5) and torch.allclose(torch.det(R), torch.ones(9), atol=1e-5)
    assert torch.allclose(T, torch.tensor([0.0, 0.0, 2.5]).expand(9, 3), atol=1e-5)  # origin -> (0, 0, dist)
    C = -torch.bmm(T[:, None], R.transpose(1, 2))[:, 0]  # camera centre in the world
    el, az = math.radians(30.0), azim * math.pi / 180
    want = 2.5 * torch.stack([math.cos(el) * torch.sin(az), torch.full_like(az, math.sin(el)), math.cos(el) * torch.cos(az)], -1)
    assert torch.allclose(C, want, atol=1e-5)
    up_view = torch.tensor([0.0, 1.0, 0.0]) @ R  # world +Y seen from the camera: positive y, no x component
    assert (up_view[:, 1] > 0.5).all() and torch.allclose(up_view[:, 0], torch.zeros(9), atol=1e-6)
    # straight down the up axis: still a proper rotation
    R2, _ = look_at_view_transform(dist=1.0, elev=90.0, azim=0.0)
    assert torch.allclose(R2 @ R2.transpose(1, 2), torch.eye(3)[None], atol=1e-4)


@pytest.mark.parametrize("H,W", [(64, 64), (48, 80), (80, 48)])
def test_transform_points_screen_inverts_the_pixel_centre_map(H, W):
    """screen (x, y) of a point = (column, row) of the pixel whose centre it projects to, for the rasterizer's map
    x_ndc = r_x (1 - (2 col + 1) / W), r_x = max(W / H, 1) (oracle: pix_to_ndc), on both camera classes"""
    s = 1.0 / math.tan(math.radians(15.0))
    for cam in (FoVPerspectiveCameras(fov=30.0), PerspectiveCameras(focal_length=s)):
        cols, rows = torch.meshgrid(torch.arange(0, W, 7.0), torch.arange(0, H, 5.0), indexing="xy")
        x_ndc = max(W / H, 1.0) * (1 - (2 * cols + 1) / W)
        y_ndc = max(H / W, 1.0) * (1 - (2 * rows + 1) / H)
        z = torch.full_like(x_ndc, 3.0)
        pts = torch.stack([x_ndc * z / s, y_ndc * z / s, z], -1).reshape(1, -1, 3)
        scr = cam.transform_points_screen(pts, image_size=(H, W))
        assert torch.allclose(scr[0, :, 0], cols.flatten(), atol=1e-3)
        assert torch.allclose(scr[0, :, 1], rows.flatten(), atol=1e-3)
        assert torch.allclose(scr[0, :, 2], z.flatten())
        # with a camera pose: same answer for the world points that map to those view points
        R = synth.pose_to_R(torch.tensor([40.0]), torch.tensor([10.0]), torch.tensor([5.0]))
        T = torch.tensor([[0.1, -0.2, 2.7]])
        world = (pts - T[:, None]) @ R.transpose(1, 2)
        scr2 = cam.transform_points_screen(world, image_size=((H, W),), R=R, T=T)
        assert torch.allclose(scr2, scr, atol=2e-3)


def test_repeat_of_equal_sized_meshes_copies_nothing_until_asked():
    """utils/mesh.py:43-71 tiles verts / faces / maps K times; here the tiling is an index (image n = mesh n % B) and
    the repeated tensors exist only for callers that ask for them"""
    from unicorn_amd.structures import TiledMeshes
    v, f = synth.icosphere(1)
    verts = torch.stack([v, 2 * v, 3 * v]).requires_grad_(True)
    maps = torch.rand(3, 4, 4, 3, requires_grad=True)
    uvs = synth.spherical_uvs(v)
    fb, ub = f[None].expand(3, -1, -1), uvs[None].expand(3, -1, -1)
    m = Meshes(verts, fb, TexturesUV(maps, fb, ub))
    r = repeat(m, 2)
    assert isinstance(r, TiledMeshes) and r.tile_base is m and r.tile_reps == 2
    assert len(r) == 6 and r.equisized and r.device == verts.device and len(r.textures) == 6
    assert isinstance(r.textures, TexturesUV) and r.textures.tile_base is m.textures
    assert r._tiled_verts is None and r.textures._tiled is None            # nothing copied so far
    # ordinary consumers see ordinary tiled tensors, autograd-connected to the base
    assert torch.equal(r.verts_padded(), verts.repeat(2, 1, 1)) and r.verts_padded().requires_grad
    assert r.faces_padded().shape == (6, f.shape[0], 3) and r.faces_padded().stride(0) == 0   # shared topology: a view
    assert torch.equal(r.textures.maps_padded(), maps.repeat(2, 1, 1, 1))
    assert r.textures.faces_uvs_padded().stride(0) == 0 and r.textures._shared_tables()
    assert torch.equal(r.verts_list()[4], verts[1]) and torch.equal(r.faces_packed()[-1], f[-1] + 5 * v.shape[0])
    r.verts_padded().sum().backward()
    assert torch.equal(verts.grad, torch.full_like(verts, 2.0))
    d = repeat(m.detach(), 2)                                             # unicorn.py:227-228,232: detach, then repeat
    assert not d.tile_base.verts_padded().requires_grad
    # an in-place change of the repeated batch ends the tiling (the copies are no longer equal to their base)
    r2 = repeat(m, 2)
    r2.offset_verts_(torch.ones(6 * v.shape[0], 3))
    assert r2.tile_base is None and torch.allclose(r2.verts_padded()[3], verts.detach()[0] + 1)
    assert r2.faces_padded().shape[0] == 6
    # per-mesh face tables are repeated, not expanded
    m3 = Meshes(verts.detach(), f[None].repeat(3, 1, 1))
    assert repeat(m3, 2).faces_padded().stride(0) != 0 and repeat(m3, 2).textures is None
    # meshes of different sizes: plain list tiling
    v2, f2 = synth.icosphere(2)
    h = repeat(Meshes([v, v2], [f, f2]), 2)
    assert not isinstance(h, TiledMeshes) and [x.shape[0] for x in h.verts_list()] == [42, 162, 42, 162]


def test_join_meshes_as_batch_keeps_textures_and_shared_tables():
    """pytorch3d's join_meshes_as_batch: batches in order, textures joined; tables that every input holds as an expanded
    view of ONE tensor (unicorn.py:262-266,283) stay such a view, so the renderer's shared-table paths survive"""
    v, f = synth.icosphere(1)
    uvs = synth.spherical_uvs(v)
    batches = []
    for seed in range(3):
        g = torch.Generator().manual_seed(seed)
        verts = (v[None] + 0.01 * torch.randn(2, *v.shape, generator=g)).requires_grad_(True)
        maps = torch.rand(2, 4, 4, 3, generator=g).requires_grad_(True)
        fb, ub = f[None].expand(2, -1, -1), uvs[None].expand(2, -1, -1)
        batches.append(Meshes(verts, fb, TexturesUV(maps, fb, ub)))
    j = join_meshes_as_batch(batches)
    assert len(j) == 6 and j.equisized
    assert torch.equal(j.verts_padded()[2:4], batches[1].verts_padded())
    assert torch.equal(j.textures.maps_padded()[4:], batches[2].textures.maps_padded())
    assert j.faces_padded().stride(0) == 0 and j.faces_padded().data_ptr() == f.data_ptr()
    assert j.textures._shared_tables() and j.textures.faces_uvs_padded().shape == (6, f.shape[0], 3)
    (j.verts_padded().sum() + j.textures.maps_padded().sum()).backward()
    assert all(torch.equal(b.verts_padded().grad, torch.ones_like(b.verts_padded())) for b in batches)
    # per-mesh tables are concatenated; without textures on request; one batch is returned as it is
    own = [Meshes(b.verts_padded().detach(), f[None].repeat(2, 1, 1), b.textures) for b in batches]
    k = join_meshes_as_batch(own, include_textures=False)
    assert k.textures is None and k.faces_padded().stride(0) != 0 and k.faces_padded().shape[0] == 6
    assert join_meshes_as_batch([batches[0]]) is batches[0]
    # vertex colours join too; mixed kinds do not
    tv = [Meshes(b.verts_padded().detach(), b.faces_padded(), TexturesVertex(torch.rand(2, v.shape[0], 3))) for b in batches]
    assert join_meshes_as_batch(tv).textures.verts_features_padded().shape == (6, v.shape[0], 3)
    with pytest.raises(ValueError):
        join_meshes_as_batch([batches[0], tv[0]])
    # meshes of different sizes: joined through the lists
    v2, f2 = synth.icosphere(2)
    h = join_meshes_as_batch([Meshes(v[None], f[None]), Meshes(v2[None], f2[None])])
    assert len(h) == 2 and not h.equisized


def test_textures_sample_their_own_texels():
    """pytorch3d's textures.sample_textures(fragments[, faces_packed]) -- what meshes.sample_textures(fragments) calls"""
    from types import SimpleNamespace
    v, f = synth.icosphere(1)
    uvs = synth.spherical_uvs(v)
    g = torch.Generator().manual_seed(3)
    N, H, W, K = 2, 5, 4, 3
    maps = torch.rand(N, 6, 6, 3, generator=g)
    fb, ub = f[None].expand(N, -1, -1), uvs[None].expand(N, -1, -1)
    m = Meshes(v[None].repeat(N, 1, 1), fb, TexturesUV(maps, fb, ub))
    p2f = torch.randint(-1, f.shape[0], (N, H, W, K), generator=g)
    p2f = torch.where(p2f >= 0, p2f + torch.arange(N).view(N, 1, 1, 1) * f.shape[0], p2f)
    bary = torch.rand(N, H, W, K, 3, generator=g)
    bary = bary / bary.sum(-1, keepdim=True)
    frags = SimpleNamespace(pix_to_face=p2f, bary_coords=bary)
    assert torch.equal(m.textures.sample_textures(frags), m.sample_textures(frags))
    mv = Meshes(m.verts_padded(), fb, TexturesVertex(torch.rand(N, v.shape[0], 3, generator=g)))
    assert torch.equal(mv.textures.sample_textures(frags, faces_packed=mv.faces_packed()), mv.sample_textures(frags))
    with pytest.raises(ValueError):
        mv.textures.sample_textures(frags)


def test_sample_textures_equals_grid_sample_with_empty_slots_at_the_origin():
    """pytorch3d's TexturesUV.sample_textures: uv interpolated with the barycentrics (0 at empty slots), then
    grid_sample(flip(maps), 2uv-1, bilinear, align_corners, border).  The implementation samples empty slots
    elsewhere and patches their value in (no contended atomics in the backward): texels and the maps' gradient must
    be those of the plain formula."""
    from types import SimpleNamespace
    from torch.nn import functional as F
    v, f = synth.icosphere(1)
    uvs = synth.spherical_uvs(v)
    g = torch.Generator().manual_seed(4)
    N, H, W, K = 2, 6, 5, 4
    fb, ub = f[None].expand(N, -1, -1), uvs[None].expand(N, -1, -1)
    p2f = torch.randint(-60, f.shape[0], (N, H, W, K), generator=g).clamp(min=-1)
    p2f = torch.where(p2f >= 0, p2f + torch.arange(N).view(N, 1, 1, 1) * f.shape[0], p2f)
    bary = torch.rand(N, H, W, K, 3, generator=g)
    bary = bary / bary.sum(-1, keepdim=True)
    cot = torch.randn(N, H, W, K, 3, generator=g)
    frags = SimpleNamespace(pix_to_face=p2f, bary_coords=bary)
    m1 = torch.rand(N, 7, 9, 3, generator=g).requires_grad_(True)
    m2 = m1.detach().clone().requires_grad_(True)
    got = Meshes(v[None].repeat(N, 1, 1), fb, TexturesUV(m1, fb, ub)).sample_textures(frags)
    fuv = uvs[f]                                                   # (F,3,2)
    uv = (bary[..., None] * fuv.repeat(N, 1, 1)[p2f.clamp(min=0)]).sum(-2).masked_fill((p2f < 0)[..., None], 0.0)
    grid = (uv * 2 - 1).permute(0, 3, 1, 2, 4).reshape(N, K * H, W, 2)
    want = F.grid_sample(torch.flip(m2.permute(0, 3, 1, 2), [2]), grid, mode="bilinear", align_corners=True,
                         padding_mode="border").view(N, 3, K, H, W).permute(0, 3, 4, 2, 1)
    assert torch.allclose(got, want, rtol=0, atol=1e-6) and int((p2f < 0).sum()) > 50
    got.backward(cot)
    want.backward(cot)
    assert torch.allclose(m1.grad, m2.grad, rtol=1e-5, atol=1e-5)


def test_vertex_textures_of_unequal_meshes():
    """pytorch3d's TexturesVertex takes a list of (V_n, C) tensors of different lengths: padded / packed / list views"""
    a, b = torch.rand(5, 3), torch.rand(3, 3)
    t = TexturesVertex([a, b])
    assert len(t) == 2 and t.verts_features_padded().shape == (2, 5, 3) and t._num_verts_per_mesh == [5, 3]
    assert torch.equal(t.verts_features_packed(), torch.cat([a, b])) and torch.equal(t.verts_features_list()[1], b)
    assert torch.equal(t.verts_features_padded()[1, 3:], torch.zeros(2, 3))
    assert torch.equal(t[1].verts_features_packed(), b) and torch.equal(t[[1, 0]].verts_features_list()[1], a)
    assert torch.equal(t.extend(2).verts_features_packed(), torch.cat([a, a, b, b]))
    assert torch.equal(t.clone().detach().verts_features_packed(), t.verts_features_packed())


def test_faces_verts_textures_packed_reads_the_texture_at_the_corners():
    """pytorch3d TexturesUV.faces_verts_textures_packed (utils/mesh.py:162-165 builds vertex colours from it): uv (u, v)
    reads map row (1 - v) * (Ht - 1), column u * (Wt - 1), bilinearly"""
    Ht, Wt = 5, 7
    maps = torch.rand(2, Ht, Wt, 3, generator=torch.Generator().manual_seed(6))
    # corners placed on texel centres, plus one half-way between two texels
    uvs = torch.tensor([[0.0, 0.0], [1.0, 1.0], [2 / 6, 3 / 4], [0.5 / 6, 0.0]])
    faces = torch.tensor([[0, 1, 2], [2, 3, 0]])
    t = TexturesUV(maps, faces[None].expand(2, -1, -1), uvs[None].expand(2, -1, -1))
    out = t.faces_verts_textures_packed()
    assert out.shape == (4, 3, 3)
    m = maps[1]
    want = {0: m[Ht - 1, 0], 1: m[0, Wt - 1], 2: m[1, 2], 3: 0.5 * (m[Ht - 1, 0] + m[Ht - 1, 1])}
    for face in range(2):
        for corner in range(3):
            assert torch.allclose(out[2 + face, corner], want[int(faces[face, corner])], atol=1e-6)
