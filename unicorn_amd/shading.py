"""`meshes.sample_textures(fragments)` for callers that want the texels themselves (N,H,W,K,C), as pytorch3d returns
them (renderer.py:162,170).  NOT on the training path: `LayeredShader` fuses sampling and blending in one HIP kernel
(ur_shade_blend) and never materialises texels.  This convenience composes stock torch ops."""
import torch
from torch.nn import functional as F

from .structures import TexturesUV, TexturesVertex


_SPARE_ROWS = 4096


def _rows_sum(idx, contrib, rows, use_atomics=None):
    """out[r] = sum of contrib[i] over idx[i] == r, r < rows.  idx < 0 marks entries that count for nothing (empty
    fragment slots -- most of a K-deep fragment tensor): they are dealt over _SPARE_ROWS rows behind the table instead
    of all meeting on one row, which is what made the obvious formulations slow at 32 x 64x64 x 25 fragments: autograd's
    index_put backward of table[idx.clamp(0)] 0.4 s, index_add_ on the clamped index 77 ms (whole waves queue on row
    0's address), a sorted segmented sum 0.4 s (row 0's segment is walked by one thread)."""
    n = idx.shape[0]
    spare = rows + (torch.arange(n, device=idx.device) & (_SPARE_ROWS - 1))
    where = torch.where(idx < 0, spare, idx)
    out = contrib.new_zeros((rows + _SPARE_ROWS,) + tuple(contrib.shape[1:])).index_add_(0, where, contrib)
    return out[:rows]


class _InterpolateFaceAttrs(torch.autograd.Function):
    """sum_c bary_c * face_attrs[face, c] with its own backward: autograd's backward of the advanced-indexing gather
    is an index_put that takes 0.4 s per call at 32 x 64x64 x 25 fragments on this stack (five such gathers per lit
    render); the per-face sums are taken by _rows_sum instead."""

    @staticmethod
    def forward(ctx, pix_to_face, bary, face_attrs):
        mask = pix_to_face < 0
        idx = pix_to_face.long().reshape(-1)                             # (< 0: empty slot)
        w = bary.masked_fill(mask[..., None], 0.0)                       # empty slots weigh nothing, both ways
        attrs = face_attrs.index_select(0, idx.clamp(min=0)).view(*pix_to_face.shape, 3, -1)
        ctx.save_for_backward(idx, w, face_attrs, mask)
        return (w[..., None] * attrs).sum(-2)

    @staticmethod
    def backward(ctx, g):
        idx, w, face_attrs, mask = ctx.saved_tensors
        g_bary = g_attrs = None
        if ctx.needs_input_grad[1]:  # (slots that hold no face get no barycentric gradient: their output is masked)
            attrs = face_attrs.index_select(0, idx.clamp(min=0)).view(*w.shape[:-1], 3, -1)
            g_bary = (g[..., None, :] * attrs).sum(-1).masked_fill(mask[..., None], 0.0)
        if ctx.needs_input_grad[2]:
            contrib = (w[..., None] * g[..., None, :]).reshape(-1, 3, g.shape[-1])
            g_attrs = _rows_sum(idx, contrib, face_attrs.shape[0])
        return None, g_bary, g_attrs


class _GatherRows(torch.autograd.Function):
    """table[idx] for a large index tensor, backward by scatter-add (see _InterpolateFaceAttrs)"""

    @staticmethod
    def forward(ctx, table, idx):
        flat = idx.reshape(-1)
        ctx.save_for_backward(flat)
        ctx.table_shape = tuple(table.shape)
        return table.index_select(0, flat.clamp(min=0)).view(*idx.shape, *table.shape[1:])

    @staticmethod
    def backward(ctx, g):
        (flat,) = ctx.saved_tensors
        return _rows_sum(flat, g.reshape(flat.shape[0], *ctx.table_shape[1:]), ctx.table_shape[0]), None


def gather_rows(table, idx):
    """table[idx] (idx any shape, int64) -> idx.shape + table.shape[1:].  idx < 0 = an empty slot: it reads row 0 (mask
    the result) and sends no gradient anywhere"""
    return _GatherRows.apply(table, idx)


def interpolate_face_attributes(pix_to_face, bary, face_attrs):
    """pix_to_face (N,H,W,K) packed, bary (N,H,W,K,3), face_attrs (NF,3,D) -> (N,H,W,K,D); 0 where empty."""
    return _InterpolateFaceAttrs.apply(pix_to_face, bary, face_attrs)


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
        uv = interpolate_face_attributes(p2f, bary, fuv)  # (N,H,W,K,2); (0, 0) at empty slots
        N, H, W, K = p2f.shape
        # Empty slots -- most of a K-deep fragment tensor -- all sample uv = (0, 0), and grid_sample's backward then adds
        # their (zero) gradients to ONE texel with float atomics: 17 ms of a 23 ms lit render at 32 x 64x64 x 25.  They
        # sample scattered positions instead and are given the value at (0, 0) afterwards: same texels, and the only
        # gradient that reaches a texel through them is the (0, 0) texel's own, by a reduction.
        empty = (p2f < 0)[..., None]
        i = torch.arange(p2f.numel(), device=p2f.device, dtype=torch.float32).view(N, H, W, K, 1)
        scattered = torch.cat([(i * 0.6180339887) % 1.0, (i * 0.7548776662) % 1.0], -1)
        uv = torch.where(empty, scattered, uv)
        maps = torch.flip(tex.maps_padded().permute(0, 3, 1, 2), [2])
        grid = (uv * 2.0 - 1.0).permute(0, 3, 1, 2, 4).reshape(N, K * H, W, 2)
        texels = F.grid_sample(maps, grid, mode="bilinear", align_corners=True, padding_mode="border")
        texels = texels.view(N, -1, K, H, W).permute(0, 3, 4, 2, 1)
        at_origin = tex.maps_padded()[:, -1, 0, :]           # uv = (0, 0): first column of the LAST row (v is flipped)
        return torch.where(empty, at_origin[:, None, None, None, :], texels)
    raise NotImplementedError(type(tex))
