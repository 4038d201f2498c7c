"""torch.autograd glue around the C ABI (unicorn_amd.native).  One Function per native stage, each honouring
ctx.needs_input_grad so that the pose step (grads to R, T only) and the 3D step (grads to verts / texture only) of
the reference's alternate optimisation (unicorn.py:226-230) skip the work they do not need."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import native


class _ProjectFaces(Function):
    """world verts -> NDC face corners (ur_project_faces); replaces MeshRasterizer.transform + the faces gather."""

    @staticmethod
    def forward(ctx, verts, faces, R, T, sx, sy):
        verts, R, T = verts.contiguous(), R.contiguous(), T.contiguous()
        ctx.save_for_backward(verts, faces, R, T)
        ctx.scale = (sx, sy)
        return native.project_faces(verts, faces, R, T, sx, sy)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_face_verts):
        verts, faces, R, T = ctx.saved_tensors
        need_v, _, need_R, need_T = ctx.needs_input_grad[:4]
        gv, gR, gT = native.project_faces_backward(verts, faces, R, T, ctx.scale[0], ctx.scale[1],
                                                   grad_face_verts.contiguous(), need_v, need_R, need_T)
        return gv, None, gR, gT, None, None


class _Rasterize(Function):
    """ur_rasterize_meshes / ur_rasterize_meshes_backward; mirrors pytorch3d's _RasterizeFaceVerts."""

    @staticmethod
    def forward(ctx, face_verts, first, nfaces, image_size, blur_radius, K, clip, cull):
        face_verts = face_verts.contiguous()
        p2f, zbuf, bary, dists = native.rasterize_meshes(face_verts, first, nfaces, image_size, blur_radius, K, clip,
                                                         cull)
        ctx.save_for_backward(face_verts, p2f)
        ctx.clip = clip
        ctx.mark_non_differentiable(p2f)
        return p2f, zbuf, bary, dists

    @staticmethod
    @once_differentiable
    def backward(ctx, _g_p2f, g_zbuf, g_bary, g_dists):
        face_verts, p2f = ctx.saved_tensors
        c = lambda g: None if g is None else g.contiguous()
        gfv = native.rasterize_meshes_backward(face_verts, p2f, c(g_zbuf), c(g_bary), c(g_dists), ctx.clip)
        return gfv, None, None, None, None, None, None, None


class _ShadeBlend(Function):
    """ur_shade_blend / ur_shade_blend_backward: TexturesUV sampling + ambient + layered_rgb_blend in one kernel."""

    @staticmethod
    def forward(ctx, bary, dists, maps, p2f, face_uvs, shared, first, sigma, background, ambient, mode):
        bary, dists, maps = bary.contiguous(), dists.contiguous(), maps.contiguous()
        ctx.save_for_backward(bary, dists, maps, p2f, face_uvs, first)
        ctx.cfg = (shared, sigma, background, ambient, mode)
        return native.shade_blend(p2f, bary, dists, maps, face_uvs, shared, first, sigma, background, ambient, mode)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_image):
        bary, dists, maps, p2f, face_uvs, first = ctx.saved_tensors
        shared, sigma, background, ambient, mode = ctx.cfg
        need_bary, need_dists, need_maps = ctx.needs_input_grad[:3]
        gmaps, gbary, gdists = native.shade_blend_backward(grad_image.contiguous(), p2f, bary, dists, maps, face_uvs,
                                                           shared, first, sigma, background, ambient, mode, need_maps)
        return (gbary if need_bary else None, gdists if need_dists else None, gmaps, None, None, None, None, None,
                None, None, None)


class _BlendColors(Function):
    """ur_blend_colors / ur_blend_colors_backward: layered or softmax blending of caller-computed fragment colours."""

    @staticmethod
    def forward(ctx, colors, dists, zbuf, p2f, kind, sigma, gamma, background, znear, zfar):
        colors, dists = colors.contiguous(), dists.contiguous()
        zbuf = None if zbuf is None else zbuf.contiguous()
        ctx.save_for_backward(colors, dists, zbuf, p2f)
        ctx.cfg = (kind, sigma, gamma, background, znear, zfar)
        return native.blend_colors(colors, p2f, dists, zbuf, kind, sigma, gamma, background, znear, zfar)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_image):
        colors, dists, zbuf, p2f = ctx.saved_tensors
        kind, sigma, gamma, background, znear, zfar = ctx.cfg
        gcol, gd, gz = native.blend_colors_backward(grad_image.contiguous(), colors, p2f, dists, zbuf, kind, sigma,
                                                    gamma, background, znear, zfar)
        need = ctx.needs_input_grad
        return (gcol if need[0] else None, gd if need[1] else None, gz if (need[2] and zbuf is not None) else None,
                None, None, None, None, None, None, None)


def blend_colors(colors, p2f, dists, zbuf, kind, sigma, gamma, background, znear=1.0, zfar=100.0):
    return _BlendColors.apply(colors, dists, zbuf, p2f, int(kind), float(sigma), float(gamma), tuple(background),
                              float(znear), float(zfar))


class _RenderLayered(Function):
    """ur_render_layered_forward / _backward: rasterize + sample + blend in one launch each way."""

    @staticmethod
    def forward(ctx, face_verts, maps, first, nfaces, image_size, blur_radius, K, clip, face_uvs, shared, sigma,
                background, ambient, mode):
        face_verts, maps = face_verts.contiguous(), maps.contiguous()
        image, layers, ws = native.render_layered_forward(face_verts, first, nfaces, image_size, blur_radius, K, clip,
                                                          maps, face_uvs, shared, sigma, background, ambient, mode)
        ctx.save_for_backward(maps, first, nfaces, face_uvs, layers, ws)
        ctx.cfg = (image_size, blur_radius, K, clip, shared, sigma, background, ambient, mode, face_verts.shape[0])
        return image

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_image):
        maps, first, nfaces, face_uvs, layers, ws = ctx.saved_tensors
        image_size, blur_radius, K, clip, shared, sigma, background, ambient, mode, NF = ctx.cfg
        need_fv, need_maps = ctx.needs_input_grad[:2]
        gfv, gmaps = native.render_layered_backward(grad_image.contiguous(), first, nfaces, NF, image_size,
                                                    blur_radius, K, clip, maps, face_uvs, shared, sigma, background,
                                                    ambient, mode, layers, ws, need_fv, need_maps)
        return (gfv, gmaps) + (None,) * 12


def render_layered(face_verts, maps, first, nfaces, image_size, blur_radius, K, clip, face_uvs, shared, sigma,
                   background, ambient, mode):
    return _RenderLayered.apply(face_verts, maps, first, nfaces, tuple(image_size), float(blur_radius), int(K),
                                bool(clip), face_uvs, bool(shared), float(sigma), tuple(background),
                                None if ambient is None else tuple(ambient), int(mode))


class _RenderMeshes(Function):
    """ur_render_meshes_forward / _backward: camera + rasterizer + shader of a batch of equal-sized meshes in one call
    each way (MeshRenderer.forward, renderer.py:48-49).  The forward also clears the backward's gradient accumulators
    (no fill launch in the backward) when it knows which gradients will be wanted.  Everything that is not a tensor
    travels as ONE tuple: autograd's per-argument bookkeeping is a visible share of an eager step's host time."""

    @staticmethod
    def forward(ctx, verts, maps, R, T, faces, first, nfaces, face_uvs, cfg):
        verts, maps, R, T = verts.contiguous(), maps.contiguous(), R.contiguous(), T.contiguous()
        sx, sy, image_size, blur_radius, K, clip, shared, sigma, background, ambient, mode = cfg
        need_v, need_m, need_R, need_T = ctx.needs_input_grad[:4]
        image, state, gmaps = native.render_meshes_forward(
            verts, faces, R, T, sx, sy, first, nfaces, image_size, blur_radius, K, clip, maps, face_uvs, shared, sigma,
            background, ambient, mode, need_geometry_grads=need_v or need_R or need_T, need_map_grads=need_m)
        ctx.save_for_backward(verts, maps, R, T)
        # the step's own buffers (never seen by the caller, so no version counting needed) and the constant index /
        # UV tables (no gradient, no grad_fn: no cycle through ctx) as plain attributes -- a SavedVariable each way
        # is ~1 us of host time per tensor
        ctx.step = (state, gmaps, True)
        ctx.tables = (faces, first, nfaces, face_uvs)
        ctx.cfg = cfg
        return image

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_image):
        verts, maps, R, T = ctx.saved_tensors
        faces, first, nfaces, face_uvs = ctx.tables
        sx, sy, image_size, blur_radius, K, clip, shared, sigma, background, ambient, mode = ctx.cfg
        need_v, need_m, need_R, need_T = ctx.needs_input_grad[:4]
        state, gmaps, fresh = ctx.step
        ctx.step = (state, None, False)  # a second backward (retain_graph) clears its own accumulators
        gv, gR, gT, gmaps = native.render_meshes_backward(
            grad_image.contiguous(), verts, faces, R, T, sx, sy, first, nfaces, image_size, blur_radius, K, clip, maps,
            face_uvs, shared, sigma, background, ambient, mode, state, gmaps, fresh, need_v, need_R, need_T, need_m)
        return gv, gmaps, gR, gT, None, None, None, None, None


def render_meshes(verts, maps, R, T, faces, sx, sy, first, nfaces, image_size, blur_radius, K, clip, face_uvs, shared,
                  sigma, background, ambient, mode):
    return _RenderMeshes.apply(verts, maps, R, T, faces, first, nfaces, face_uvs,
                               (float(sx), float(sy), tuple(image_size), float(blur_radius), int(K), bool(clip),
                                bool(shared), float(sigma), tuple(background),
                                None if ambient is None else tuple(ambient), int(mode)))


def project_faces(verts, faces, R, T, sx, sy):
    return _ProjectFaces.apply(verts, faces, R, T, float(sx), float(sy))


def rasterize(face_verts, first, nfaces, image_size, blur_radius, K, clip, cull=False):
    return _Rasterize.apply(face_verts, first, nfaces, tuple(image_size), float(blur_radius), int(K), bool(clip),
                            bool(cull))


def shade_blend(p2f, bary, dists, maps, face_uvs, shared, first, sigma, background, ambient, mode):
    return _ShadeBlend.apply(bary, dists, maps, p2f, face_uvs, bool(shared), first, float(sigma), tuple(background),
                             None if ambient is None else tuple(ambient), int(mode))
