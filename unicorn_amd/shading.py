"""`meshes.sample_textures(fragments)` for callers that want the texels themselves (N,H,W,K,C), as pytorch3d returns
them (renderer.py:162,170).  NOT on the training path: `LayeredShader` fuses sampling and blending in one HIP kernel
(ur_shade_blend) and never materialises texels.  This convenience composes stock torch ops."""
import torch
from torch.nn import functional as F

from .structures import TexturesUV, TexturesVertex


def interpolate_face_attributes(pix_to_face, bary, face_attrs):
    """pix_to_face (N,H,W,K) packed, bary (N,H,W,K,3), face_attrs (NF,3,D) -> (N,H,W,K,D); 0 where empty."""
    mask = pix_to_face < 0
    idx = pix_to_face.clamp(min=0).long()
    attrs = face_attrs[idx]  # (N,H,W,K,3,D)
    out = (bary[..., None] * attrs).sum(-2)
    return out.masked_fill(mask[..., None], 0.0)


def sample_textures(meshes, fragments):
    tex = meshes.textures
    p2f, bary = fragments.pix_to_face, fragments.bary_coords
    if isinstance(tex, TexturesVertex):
        feats = tex.verts_features_packed()[meshes.faces_packed()]
        return interpolate_face_attributes(p2f, bary, feats)
    if isinstance(tex, TexturesUV):
        fuv, shared = tex.faces_verts_uvs()
        if shared:
            fuv = fuv.repeat(len(meshes), 1, 1)
        uv = interpolate_face_attributes(p2f, bary, fuv)  # (N,H,W,K,2)
        N, H, W, K = p2f.shape
        maps = torch.flip(tex.maps_padded().permute(0, 3, 1, 2), [2])
        grid = (uv * 2.0 - 1.0).permute(0, 3, 1, 2, 4).reshape(N, K * H, W, 2)
        texels = F.grid_sample(maps, grid, mode="bilinear", align_corners=True, padding_mode="border")
        return texels.view(N, -1, K, H, W).permute(0, 3, 4, 2, 1)
    raise NotImplementedError(type(tex))
