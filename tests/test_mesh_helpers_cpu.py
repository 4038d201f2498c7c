"""The reference's mesh helpers (src/utils/mesh.py) on this package's Meshes: CPU checks against plain formulas."""
import math

import torch

from unicorn_amd import mesh as M
from unicorn_amd.structures import Meshes, TexturesUV


def _two_meshes():
    v3, f3 = M.ico_sphere(1).get_mesh_verts_faces(0)
    v2, f2 = M.ico_sphere(0).get_mesh_verts_faces(0)
    return Meshes([v3 * torch.tensor([1.0, 0.5, 0.25]) + torch.tensor([0.3, -0.2, 1.0]), v2 * 3.0 - 1.0], [f3, f2])


def test_ico_sphere_counts_winding_and_closedness():
    for level, (nv, nf) in enumerate([(12, 20), (42, 80), (162, 320), (642, 1280)]):
        v, f = M.ico_sphere(level).get_mesh_verts_faces(0)
        assert v.shape == (nv, 3) and f.shape == (nf, 3)
        assert torch.allclose(v.norm(dim=1), torch.ones(nv), atol=1e-6)
        tri = v[f]
        normal = torch.linalg.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
        assert ((normal * tri.mean(1)).sum(1) > 0).all()                                 # outward winding
        e = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]).sort(dim=1)[0]
        assert (torch.unique(e, dim=0, return_counts=True)[1] == 2).all()                 # closed 2-manifold
        assert nv - len(torch.unique(e, dim=0)) + nf == 2                                # Euler characteristic of a sphere


def test_get_icosphere_orders_vertices_and_colours_them():
    plain = M.get_icosphere(2)
    mesh = M.get_icosphere(2, order_verts_by=1, colored=True)
    v, f = mesh.get_mesh_verts_faces(0)
    assert (v[1:, 1] >= v[:-1, 1]).all()
    pv, pf = plain.get_mesh_verts_faces(0)
    def rows(t):                                                                          # same triangles, renumbered
        return sorted(map(tuple, t.numpy().reshape(-1, 9).tolist()))
    assert rows(pv[pf]) == rows(v[f])
    c = mesh.textures.verts_features_padded()[0]
    assert c.shape == v.shape and float(c.min()) == 0.0 and float(c.max()) == 1.0
    assert torch.allclose(c, (v - v.min(0)[0]) / (v.max(0)[0] - v.min(0)[0]))


def test_normalize_centres_and_scales_ragged_batches():
    meshes = _two_meshes()
    for mode, measure in (("unit_cube", lambda v: float(v.abs().max()) * 2), ("unit_sphere", lambda v: float(v.norm(dim=1).max()))):
        out = M.normalize(meshes, scale_mode=mode)
        for n in range(2):
            v = out.get_mesh_verts_faces(n)[0]
            assert torch.allclose(0.5 * (v.max(0)[0] + v.min(0)[0]), torch.zeros(3), atol=1e-6)
            assert abs(measure(v) - 1.0) < 1e-5
    kept = M.normalize(meshes, center=False, scale_mode=None)
    assert torch.equal(kept.verts_packed(), meshes.verts_packed())
    same = meshes.clone()
    assert M.normalize(same, inplace=True) is same
    assert torch.allclose(same.verts_packed(), M.normalize(meshes).verts_packed())


def test_surface_samples_lie_on_the_faces_with_area_weighting():
    g = torch.Generator().manual_seed(0)
    v = torch.tensor([[0.0, 0, 0], [1, 0, 0], [0, 1, 0], [3, 0, 0], [3, 3, 0], [0, 0, 0]])
    f = torch.tensor([[0, 1, 2], [1, 3, 4]])                                              # areas 0.5 and 3.0
    v.requires_grad_(True)
    pts = M.sample_points_from_meshes(Meshes(v[None], f[None]), 20000, generator=g)
    assert pts.shape == (1, 20000, 3) and float(pts[..., 2].detach().abs().max()) == 0.0
    in_small = (pts[0, :, 0] + pts[0, :, 1] <= 1.0 + 1e-6) & (pts[0, :, 0] <= 1.0 + 1e-6)
    assert abs(float(in_small.float().mean()) - 0.5 / 3.5) < 0.01
    small = pts[0][in_small].mean(0)
    assert torch.allclose(small, torch.tensor([1 / 3, 1 / 3, 0.0]), atol=0.01)             # uniform within a face
    pts.sum().backward()
    assert float(v.grad.abs().sum()) > 0
    centred = M.normalize(Meshes(v.detach()[None], f[None]), scale_mode=None, use_center_mass=True)
    assert centred.verts_padded().shape == (1, 6, 3)


def test_uv_maps_become_vertex_colours():
    v, f = M.ico_sphere(1).get_mesh_verts_faces(0)
    uv = torch.stack([(torch.atan2(v[:, 0], v[:, 2]) + math.pi) / (2 * math.pi), torch.acos(v[:, 1].clamp(-1, 1)) / math.pi], 1)
    ramp = torch.linspace(0, 1, 32)
    maps = torch.stack([ramp[None].expand(32, 32), ramp[:, None].expand(32, 32), torch.full((32, 32), 0.25)], dim=-1)
    B = 2
    meshes = Meshes(v[None].expand(B, -1, -1), f[None].expand(B, -1, -1))
    tex = TexturesUV(torch.stack([maps, 1 - maps]), f[None].expand(B, -1, -1), uv[None].expand(B, -1, -1))
    out = M.convert_textures_uv_to_vertex(tex, meshes)
    cols = out.verts_features_padded()
    assert cols.shape == (B, 42, 3)
    # the maps are linear ramps, so the bilinear sample at a vertex's uv is the ramp value: u along x, 1 - v along y
    want = torch.stack([uv[:, 0], 1 - uv[:, 1], torch.full_like(uv[:, 0], 0.25)], 1)
    assert torch.allclose(cols[0], want, atol=1e-5) and torch.allclose(cols[1], 1 - want, atol=1e-5)
