"""The mesh helpers the reference keeps beside its renderer (src/utils/mesh.py), under the reference's names, on this
package's own `Meshes` / textures: the callers on the input side of `Renderer.forward`.

* normalize                       utils/mesh.py:21-40   centre + rescale a batch (evaluation, unicorn.py:560-561)
* repeat                          utils/mesh.py:43-71   pose tiling; here an index, not copies (structures.repeat)
* get_icosphere / ico_sphere      utils/mesh.py:74-89   the template every batch is extended from (unicorn.py:86-92)
* normal_consistency              utils/mesh.py:92-139  -> regularizers.normal_consistency (one fused launch)
* convert_textures_uv_to_vertex   utils/mesh.py:162-165 UV maps -> per-vertex colours (before an .obj is written)
* sample_points_from_meshes       pytorch3d.ops         area-weighted surface samples (normalize's centre of mass)

File output (`save_mesh_as_obj`, `save_obj`) stays outside: section 8 marks I/O out of scope.

Parity: `ico_sphere` produces the same sphere as PyTorch3D's (same 12 + subdivided vertices on the unit sphere, same
face count, outward winding), but the ORDER of its vertices and faces is this library's own: PyTorch3D's source is not
in the container, so its order cannot be pinned ("parity unpinned").  Nothing on the rendering path depends on that
order (the deformation and texture networks are functions of vertex position, unicorn.py:259-283), and
`order_verts_by` sorts the vertices anyway."""
import torch

from . import synth
from .regularizers import laplacian_smoothing, normal_consistency  # noqa: F401  (offered under the reference's names)
from .structures import Meshes, TexturesUV, TexturesVertex, join_meshes_as_batch, repeat  # noqa: F401

EPS = 1e-4


def ico_sphere(level=0, device=None):
    """One-mesh batch holding the unit icosphere of the given subdivision level (pytorch3d.utils.ico_sphere)."""
    if level < 0:
        raise ValueError("level must be >= 0")
    verts, faces = synth.icosphere(level)
    if device is not None:
        verts, faces = verts.to(device), faces.to(device)
    return Meshes(verts[None], faces[None])


def get_icosphere(level=3, order_verts_by=None, colored=False):
    """utils/mesh.py:74-89: the icosphere, optionally with its vertices sorted along one axis (stable, like the
    reference's `sorted`) and with position-as-colour vertex textures."""
    verts, faces = synth.icosphere(level)
    if order_verts_by is not None:
        assert isinstance(order_verts_by, int)
        indices = torch.sort(verts[:, order_verts_by], stable=True)[1]
        mapping = torch.empty_like(indices)
        mapping[indices] = torch.arange(len(verts))
        verts, faces = verts[indices], mapping[faces]
    mesh = Meshes(verts[None], faces[None])
    if colored:
        lo, hi = verts.min(0)[0], verts.max(0)[0]
        mesh.textures = TexturesVertex(verts_features=((verts - lo) / (hi - lo))[None])
    return mesh


def sample_points_from_meshes(meshes, num_samples=10000, generator=None):
    """(N, num_samples, 3) points drawn uniformly over the surface of each mesh: a face by its area, then a point in it
    by the square-root barycentric map.  Differentiable with respect to the vertices."""
    out = []
    for n in range(len(meshes)):
        v, f = meshes.get_mesh_verts_faces(n)
        tri = v[f]                                                                     # (F, 3, 3)
        areas = 0.5 * torch.linalg.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]).norm(dim=1)
        pick = torch.multinomial(areas.detach().clamp(min=0) + 1e-30, num_samples, replacement=True, generator=generator)
        u = torch.rand(num_samples, 2, device=v.device, dtype=v.dtype, generator=generator)
        r = u[:, 0].sqrt()
        w = torch.stack([1 - r, r * (1 - u[:, 1]), r * u[:, 1]], dim=1)                 # (S, 3)
        out.append((w[:, :, None] * tri[pick]).sum(1))
    return torch.stack(out) if out else torch.zeros(0, num_samples, 3)


def normalize(meshes, center=True, scale_mode='unit_cube', inplace=False, use_center_mass=False):
    """utils/mesh.py:21-40: move each mesh's centre (middle of its bounding box, or the mean of 100000 surface samples)
    to the origin, then scale it into [-0.5, 0.5]^3 ('unit_cube') or the unit ball ('unit_sphere')."""
    if center:
        if use_center_mass:
            offsets = sample_points_from_meshes(meshes, 100000).mean(1)
        else:
            vp = meshes.verts_padded()
            offsets = 0.5 * (vp.max(1)[0] + vp.min(1)[0])
        offsets = offsets.repeat_interleave(meshes.num_verts_per_mesh().to(offsets.device), dim=0)  # packed
        meshes = meshes.offset_verts_(-offsets) if inplace else meshes.offset_verts(-offsets)
    if scale_mode == 'none' or scale_mode is None:
        scales = 1.
    elif scale_mode == 'unit_cube':
        scales = meshes.verts_padded().abs().flatten(1).max(1)[0] * 2
    elif scale_mode == 'unit_sphere':
        scales = meshes.verts_padded().norm(dim=2).max(1)[0]
    else:
        raise NotImplementedError
    return meshes.scale_verts_(1 / scales) if inplace else meshes.scale_verts(1 / scales)


def convert_textures_uv_to_vertex(textures_uv, meshes):
    """utils/mesh.py:162-165: per-vertex colours read from the UV maps at each face corner (a vertex shared by several
    faces keeps one of the corner values, as in the reference's indexed assignment)."""
    colors = torch.zeros_like(meshes.verts_packed())
    colors[meshes.faces_packed()] = textures_uv.faces_verts_textures_packed()
    return TexturesVertex(list(colors.split(meshes.num_verts_per_mesh().tolist())))
