"""Lighting for the non-default shading branches: `shading_type` in {phong, flat, gouraud} of LayeredShader
(renderer.py:136-141,165-171) and SoftPhongShaderPlus (renderer.py:205-208), used by the `eye_light` rotating-view GIFs
(renderer.py:225-234).  The light models restate the reference's own `diffuse` / `specular`
(pytorch3d_monkey.py:73-198) and `_apply_lighting` (:34-70); the shading functions restate pytorch3d 0.6.2's
phong_shading / gouraud_shading / flat_shading (third-party, absent here: parity unpinned, checked against a numpy
restatement in tests/test_lighting_cpu.py, gouraud end to end in tests/test_gpu_vertex_textures.py); `diffuse` and
`specular` themselves are pinned by outputs of the reference's functions (tests/golden/lighting.npz).

Plain differentiable torch ops on the fragments the HIP rasterizer produced: this is per-fragment arithmetic on a
visualisation path, the blending that follows is a HIP kernel (ur_blend_colors)."""
import torch
from torch.nn import functional as F


def _t(x, device):
    """a light / material property as (1|N, 3): the reference views whatever it is given as (-1, ..., 3)
    (pytorch3d_monkey.py:99-109), e.g. the (B, 1, 3) per-view directions of the eye-light GIFs (renderer.py:249-250)"""
    t = torch.as_tensor(x, dtype=torch.float32, device=device)
    return t if t.dim() == 2 else t.reshape(-1, 3)


class Materials:
    def __init__(self, ambient_color=((1, 1, 1),), diffuse_color=((1, 1, 1),), specular_color=((1, 1, 1),),
                 shininess=64, device="cpu"):
        self.ambient_color, self.diffuse_color, self.specular_color = ambient_color, diffuse_color, specular_color
        self.shininess, self.device = shininess, device

    def to(self, device):
        self.device = device
        return self


def _bcast(v, like):
    """(1|N, 3) light property -> broadcastable against `like` (N, ..., 3)"""
    return v.view((v.shape[0],) + (1,) * (like.dim() - 2) + (3,))


def _unit(v):
    """v / max(|v|, 1e-6) along the last axis (what the reference's normalize(p=2, eps=1e-6) computes)"""
    return v / v.norm(p=2, dim=-1, keepdim=True).clamp_min(1e-6)


def diffuse(normals, color, direction):
    """Lambert's cosine law: colour * max(n . l, 0) with n, l renormalised (pytorch3d_monkey.py:73-123).  Arguments
    broadcastable against `normals` (N, ..., 3).  Pinned by tests/golden/lighting.npz (outputs of the reference's own
    function)."""
    lambert = torch.relu((_unit(normals) * _unit(direction)).sum(-1, keepdim=True))  # relu: zero slope AT zero
    return color * lambert


def specular(points, normals, direction, color, camera_position, shininess):
    """colour * max(v . r, 0)^shininess on the lit side of the surface, r = the light direction mirrored at the
    normal, v = the unit vector towards the camera (pytorch3d_monkey.py:126-198).  Pinned by tests/golden/lighting.npz."""
    n, l = _unit(normals), _unit(direction)
    n_dot_l = (n * l).sum(-1, keepdim=True)
    mirrored = 2 * (n_dot_l * n) - l
    towards_eye = _unit(camera_position - points)
    lit_side = (n_dot_l > 0).to(points.dtype)
    highlight = torch.relu((towards_eye * mirrored).sum(-1, keepdim=True)) * lit_side
    shininess = torch.as_tensor(shininess, dtype=points.dtype, device=points.device)
    if shininess.dim() == 1:      # one exponent per batch element
        shininess = shininess.view((-1,) + (1,) * (points.dim() - 1))
    elif shininess.dim() > 1:     # one per point
        shininess = shininess.unsqueeze(-1)
    return color * highlight.pow(shininess)


def light_terms(lights, materials, points, normals, camera_center):
    """_apply_lighting (pytorch3d_monkey.py:34-70) -> ambient (N,1..,3), diffuse, specular (like points).
    points / normals: (N, ..., 3) in world space; camera_center (N, 3)."""
    dev = points.device
    amb = _bcast(_t(materials.ambient_color, dev) * _t(lights.ambient_color, dev), points)
    if not hasattr(lights, "diffuse_color"):  # AmbientLights: no diffuse / specular term (pytorch3d_monkey.py:22-26)
        return amb, torch.zeros_like(points), torch.zeros_like(points)
    if hasattr(lights, "location"):  # PointLights: direction from the surface point to the light
        direction = _bcast(_t(lights.location, dev), points) - points
    else:
        direction = _bcast(_t(lights.direction, dev), points).expand_as(points)
    dif = _bcast(_t(materials.diffuse_color, dev), points) * diffuse(normals, _bcast(_t(lights.diffuse_color, dev),
                                                                                     points), direction)
    shin = torch.as_tensor(materials.shininess, dtype=torch.float32, device=dev)
    spe = _bcast(_t(materials.specular_color, dev), points) * specular(
        points, normals, direction, _bcast(_t(lights.specular_color, dev), points),
        camera_center.view((camera_center.shape[0],) + (1,) * (points.dim() - 2) + (3,)), shin)
    return amb, dif, spe


def camera_center(R, T):
    """world position of the camera of the X_view = X_world @ R + T convention: C = -T @ R^T"""
    return -torch.bmm(T[:, None], R.transpose(1, 2))[:, 0]


def _interp(fragments, face_attrs):
    from .shading import interpolate_face_attributes
    return interpolate_face_attributes(fragments.pix_to_face, fragments.bary_coords, face_attrs)


def phong_shading(meshes, fragments, lights, materials, texels, R, T):
    """per-pixel lighting with interpolated vertex normals: colors = (ambient + diffuse) * texels + specular"""
    faces = meshes.faces_packed()
    coords = _interp(fragments, meshes.verts_packed()[faces])
    normals = _interp(fragments, meshes.verts_normals_packed()[faces])
    amb, dif, spe = light_terms(lights, materials, coords, normals, camera_center(R, T))
    return (amb + dif) * texels + spe


def flat_shading(meshes, fragments, lights, materials, texels, R, T):
    """one normal and one lighting point (the face centre) per face"""
    faces = meshes.faces_packed()
    fv = meshes.verts_packed()[faces]                      # (NF,3,3)
    centers, fnorm = fv.mean(1), meshes.faces_normals_packed()
    p2f = fragments.pix_to_face
    mask = (p2f < 0)[..., None]
    from .shading import gather_rows
    coords = gather_rows(centers, p2f.long()).masked_fill(mask, 0.0)
    normals = gather_rows(fnorm, p2f.long()).masked_fill(mask, 0.0)
    amb, dif, spe = light_terms(lights, materials, coords, normals, camera_center(R, T))
    return (amb + dif) * texels + spe


def gouraud_shading(meshes, fragments, lights, materials, R, T):
    """lighting at the vertices (vertex colours = TexturesVertex features), interpolated over the face"""
    verts, vnorm = meshes.verts_packed(), meshes.verts_normals_packed()
    vidx = meshes.verts_packed_to_mesh_idx()
    vcol = meshes.textures.verts_features_packed()
    cc = camera_center(R, T)[vidx]
    pick = lambda v: _t(v, verts.device)[vidx] if _t(v, verts.device).shape[0] > 1 else _t(v, verts.device)
    amb = pick(materials.ambient_color) * pick(lights.ambient_color)
    if hasattr(lights, "diffuse_color"):
        direction = (pick(lights.location) - verts) if hasattr(lights, "location") else pick(lights.direction).expand_as(verts)
        dif = pick(materials.diffuse_color) * diffuse(vnorm, pick(lights.diffuse_color), direction)
        shin = torch.as_tensor(materials.shininess, dtype=torch.float32, device=verts.device)
        spe = pick(materials.specular_color) * specular(verts, vnorm, direction, pick(lights.specular_color), cc, shin)
    else:
        dif, spe = torch.zeros_like(verts), torch.zeros_like(verts)
    lit = vcol * (amb + dif) + spe
    return _interp(fragments, lit[meshes.faces_packed()])
