"""Deterministic synthetic inputs for tests and bench.py (no datasets offline): icosphere meshes, spherical UVs,
poses.  Shapes and ranges follow the reference's configs:

* mesh          icosphere level L scaled to an ellipse [1, 0.7, 0.7]*s        unicorn.py:32,86-92, utils/mesh.py:74-75
* UVs           spherical map of the template vertices                        tools.py:164-171, unicorn.py:93
* rotations     R = R_azim @ R_elev @ R_roll                                   tools.py:197-231, unicorn.py:306
* translation   (0, 0, 2.732) + U(+-range)                                     unicorn.py:33, configs/car.yml:29
"""
import math

import numpy as np
import torch

SCALE_ELLIPSE = (1.0, 0.7, 0.7)
PRIOR_TRANSLATION = (0.0, 0.0, 2.732)

_ICO_CACHE = {}


def icosphere(level=3):
    """Unit icosphere: level 3 -> 642 verts / 1280 faces, 4 -> 2562 / 5120, 5 -> 10242 / 20480.
    Returns (verts float32 (V,3), faces int64 (F,3)), counter-clockwise seen from outside."""
    if level in _ICO_CACHE:
        v, f = _ICO_CACHE[level]
        return v.clone(), f.clone()
    t = (1.0 + math.sqrt(5.0)) / 2.0
    verts = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                      [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    verts /= np.linalg.norm(verts, axis=1, keepdims=True)
    faces = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                      [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
                      [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    for _ in range(level):
        edge_mid = {}
        vlist = [v for v in verts]

        def mid(a, b):
            key = (a, b) if a < b else (b, a)
            if key not in edge_mid:
                m = vlist[a] + vlist[b]
                vlist.append(m / np.linalg.norm(m))
                edge_mid[key] = len(vlist) - 1
            return edge_mid[key]

        new_faces = []
        for a, b, c in faces:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            new_faces += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        verts, faces = np.stack(vlist), np.array(new_faces, dtype=np.int64)
    out = (torch.from_numpy(verts.astype(np.float32)), torch.from_numpy(faces))
    _ICO_CACHE[level] = out
    return out[0].clone(), out[1].clone()


def spherical_uvs(X, eps=1e-5):
    """3D points -> UV in [0,1]: v = inclination from +Y / pi, u = (azimuth from +Z + pi) / 2pi."""
    radius = torch.linalg.norm(X, dim=-1).clamp(min=eps)
    theta = torch.acos((X[..., 1] / radius).clamp(min=-1 + eps, max=1 - eps))
    phi = torch.atan2(X[..., 0], X[..., 2])
    return torch.stack([(phi + math.pi) / (2 * math.pi), theta / math.pi], dim=-1)


def _deg(a):
    return a * (math.pi / 180.0)


def azim_rotation(azim):
    """(B,) degrees -> (B,3,3); angle with +X in the XZ plane."""
    c, s, z, o = torch.cos(_deg(azim)), torch.sin(_deg(azim)), torch.zeros_like(azim), torch.ones_like(azim)
    return torch.stack([c, z, s, z, o, z, -s, z, c], dim=-1).view(-1, 3, 3)


def elev_rotation(elev):
    """(B,) degrees -> (B,3,3); angle with +Z in the YZ plane."""
    c, s, z, o = torch.cos(_deg(-elev)), torch.sin(_deg(-elev)), torch.zeros_like(elev), torch.ones_like(elev)
    return torch.stack([o, z, z, z, c, s, z, -s, c], dim=-1).view(-1, 3, 3)


def roll_rotation(roll):
    """(B,) degrees -> (B,3,3); angle with +X in the XY plane."""
    c, s, z, o = torch.cos(_deg(roll)), torch.sin(_deg(roll)), torch.zeros_like(roll), torch.ones_like(roll)
    return torch.stack([c, s, z, -s, c, z, z, z, o], dim=-1).view(-1, 3, 3)


def pose_to_R(azim, elev, roll):
    return azim_rotation(azim) @ elev_rotation(elev) @ roll_rotation(roll)


def make_scene(N, level=3, tex_size=64, scale=0.6, jitter=0.02, elev_range=(-10.0, 30.0), roll_range=(0.0, 0.0),
               t_range=0.3, seed=0, device="cpu"):
    """Synthetic batch: verts (N,V,3), faces (F,3) int64, verts_uvs (V,2), maps (N,Ht,Wt,3) in (0,1), R (N,3,3),
    T (N,3).  Seeds: +1 vertex jitter, +2 textures, +3 poses."""
    v0, faces = icosphere(level)
    v0 = v0 * (torch.tensor(SCALE_ELLIPSE) * scale)
    uvs = spherical_uvs(v0)
    g = torch.Generator().manual_seed(seed + 1)
    verts = v0[None] + jitter * torch.randn(N, v0.shape[0], 3, generator=g)
    g = torch.Generator().manual_seed(seed + 2)
    maps = torch.rand(N, tex_size, tex_size, 3, generator=g)
    g = torch.Generator().manual_seed(seed + 3)
    azim = torch.rand(N, generator=g) * 360.0
    elev = elev_range[0] + torch.rand(N, generator=g) * (elev_range[1] - elev_range[0])
    roll = roll_range[0] + torch.rand(N, generator=g) * (roll_range[1] - roll_range[0])
    R = pose_to_R(azim, elev, roll)
    T = torch.tensor(PRIOR_TRANSLATION)[None] + (torch.rand(N, 3, generator=g) * 2 - 1) * t_range
    out = dict(verts=verts, faces=faces, verts_uvs=uvs, maps=maps, R=R, T=T)
    return {k: v.to(device) for k, v in out.items()}
