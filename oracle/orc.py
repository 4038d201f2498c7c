"""ctypes/numpy front-end of the CPU oracle (oracle/unicorn_oracle.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (unicorn_amd/) never imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liborc.so")
_lib = None


def build(force=False):
    """Compile liborc.so with g++ (a couple of seconds)."""
    src = os.path.join(_HERE, "unicorn_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liborc.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _sfx(dtype):
    return {np.dtype(np.float32): "f32", np.dtype(np.float64): "f64"}[np.dtype(dtype)]


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _real(dtype, v):
    return ctypes.c_float(v) if np.dtype(dtype) == np.float32 else ctypes.c_double(v)


def blur_radius_for(sigma):
    """renderer.py:46 -- np.log(1. / 1e-4 - 1.) * sigma"""
    return float(np.log(1. / 1e-4 - 1.) * sigma)


def project_verts(verts, R, T, s, dtype=np.float32):
    verts, R, T = _c(verts, dtype), _c(R, dtype), _c(T, dtype)
    N, V = verts.shape[:2]
    out = np.empty_like(verts)
    getattr(lib(), "orc_project_verts_" + _sfx(dtype))(_p(verts), _p(R), _p(T), _real(dtype, s), N, V, _p(out))
    return out


def project_verts_bwd(verts, R, T, s, g_ndc, dtype=np.float32):
    verts, R, T, g_ndc = _c(verts, dtype), _c(R, dtype), _c(T, dtype), _c(g_ndc, dtype)
    N, V = verts.shape[:2]
    gv, gR, gT = np.empty_like(verts), np.empty_like(R), np.empty_like(T)
    getattr(lib(), "orc_project_verts_bwd_" + _sfx(dtype))(_p(verts), _p(R), _p(T), _real(dtype, s), N, V, _p(g_ndc),
                                                          _p(gv), _p(gR), _p(gT))
    return gv, gR, gT


def _mesh_idx(first_idx, num_faces, N, NF):
    if first_idx is None:
        assert NF % N == 0
        F = NF // N
        first_idx = np.arange(N, dtype=np.int64) * F
        num_faces = np.full(N, F, dtype=np.int64)
    return _c(first_idx, np.int64), _c(num_faces, np.int64)


def rasterize(face_verts, N, H, W, blur_radius, K, clip_bary=None, cull_backfaces=False, first_idx=None,
              num_faces=None, dtype=np.float32):
    """face_verts (NF,3,3) packed -> pix_to_face i64 (N,H,W,K), zbuf, bary (...,3), dists"""
    fv = _c(face_verts, dtype).reshape(-1, 9)
    first_idx, num_faces = _mesh_idx(first_idx, num_faces, N, fv.shape[0])
    if clip_bary is None:
        clip_bary = blur_radius > 0
    p2f = np.empty((N, H, W, K), np.int64)
    zb, ds = np.empty((N, H, W, K), dtype), np.empty((N, H, W, K), dtype)
    bar = np.empty((N, H, W, K, 3), dtype)
    getattr(lib(), "orc_rasterize_naive_" + _sfx(dtype))(_p(fv), _p(first_idx), _p(num_faces), N, H, W,
                                                        _real(dtype, blur_radius), K, int(clip_bary),
                                                        int(cull_backfaces), _p(p2f), _p(zb), _p(bar), _p(ds))
    return p2f, zb, bar, ds


def rasterize_backward(face_verts, pix_to_face, grad_zbuf, grad_bary, grad_dists, clip_bary, dtype=np.float32):
    fv = _c(face_verts, dtype).reshape(-1, 9)
    p2f = _c(pix_to_face, np.int64)
    N, H, W, K = p2f.shape
    gz, gb, gd = _c(grad_zbuf, dtype), _c(grad_bary, dtype), _c(grad_dists, dtype)
    gfv = np.zeros_like(fv)
    getattr(lib(), "orc_rasterize_backward_" + _sfx(dtype))(_p(fv), _p(p2f), _p(gz), _p(gb), _p(gd), N, H, W, K,
                                                           int(clip_bary), _p(gfv))
    return gfv.reshape(-1, 3, 3)


def tile_faces(face_verts, N, H, W, blur_radius, ts, cap, cull_backfaces=False, first_idx=None, num_faces=None,
               dtype=np.float32):
    fv = _c(face_verts, dtype).reshape(-1, 9)
    first_idx, num_faces = _mesh_idx(first_idx, num_faces, N, fv.shape[0])
    TH, TW = (H + ts - 1) // ts, (W + ts - 1) // ts
    counts = np.empty((N, TH, TW), np.int32)
    lists = np.empty((N, TH, TW, cap), np.int32)
    getattr(lib(), "orc_tile_faces_" + _sfx(dtype))(_p(fv), _p(first_idx), _p(num_faces), N, H, W,
                                                   _real(dtype, blur_radius), ts, int(cull_backfaces), cap, _p(counts),
                                                   _p(lists))
    return counts, lists


def sample_textures(maps, face_uvs, pix_to_face, bary, dtype=np.float32):
    """maps (N,Ht,Wt,C), face_uvs (NF,3,2) -> texels (N,H,W,K,C)"""
    maps, fuv, p2f, bar = _c(maps, dtype), _c(face_uvs, dtype), _c(pix_to_face, np.int64), _c(bary, dtype)
    N, H, W, K = p2f.shape
    _, Ht, Wt, C = maps.shape
    tex = np.empty((N, H, W, K, C), dtype)
    getattr(lib(), "orc_sample_textures_fwd_" + _sfx(dtype))(_p(maps), _p(fuv), _p(p2f), _p(bar), N, H, W, K, Ht, Wt, C,
                                                            _p(tex))
    return tex


def sample_textures_bwd(maps, face_uvs, pix_to_face, bary, grad_texels, dtype=np.float32):
    maps, fuv, p2f, bar = _c(maps, dtype), _c(face_uvs, dtype), _c(pix_to_face, np.int64), _c(bary, dtype)
    gtex = _c(grad_texels, dtype)
    N, H, W, K = p2f.shape
    _, Ht, Wt, C = maps.shape
    gmaps, gbar = np.zeros_like(maps), np.zeros_like(bar)
    getattr(lib(), "orc_sample_textures_bwd_" + _sfx(dtype))(_p(maps), _p(fuv), _p(p2f), _p(bar), _p(gtex), N, H, W, K,
                                                            Ht, Wt, C, _p(gmaps), _p(gbar))
    return gmaps, gbar


BLEND_HARD, BLEND_CLIP_INSIDE, BLEND_SIGMOID = 0, 1, 2


def blend_mode(sigma, clip_inside):
    """renderer.py:186-191"""
    return BLEND_HARD if sigma == 0 else (BLEND_CLIP_INSIDE if clip_inside else BLEND_SIGMOID)


def layered_blend(colors, pix_to_face, dists, sigma, background, mode, dtype=np.float32):
    col, p2f, ds, bg = _c(colors, dtype), _c(pix_to_face, np.int64), _c(dists, dtype), _c(background, dtype)
    N, H, W, K = p2f.shape
    img = np.empty((N, 4, H, W), dtype)
    getattr(lib(), "orc_layered_blend_fwd_" + _sfx(dtype))(_p(col), _p(p2f), _p(ds), N, H, W, K, _real(dtype, sigma),
                                                          _p(bg), mode, _p(img))
    return img


def layered_blend_bwd(colors, pix_to_face, dists, grad_image, sigma, background, mode, dtype=np.float32):
    col, p2f, ds, bg = _c(colors, dtype), _c(pix_to_face, np.int64), _c(dists, dtype), _c(background, dtype)
    gimg = _c(grad_image, dtype)
    N, H, W, K = p2f.shape
    gcol, gds = np.empty_like(col), np.empty_like(ds)
    getattr(lib(), "orc_layered_blend_bwd_" + _sfx(dtype))(_p(col), _p(p2f), _p(ds), _p(gimg), N, H, W, K,
                                                          _real(dtype, sigma), _p(bg), mode, _p(gcol), _p(gds))
    return gcol, gds


def softmax_blend(colors, pix_to_face, dists, zbuf, sigma, gamma, background, znear=1.0, zfar=100.0,
                  dtype=np.float32):
    col, p2f, ds, zb = _c(colors, dtype), _c(pix_to_face, np.int64), _c(dists, dtype), _c(zbuf, dtype)
    bg = _c(background, dtype)
    N, H, W, K = p2f.shape
    img = np.empty((N, 4, H, W), dtype)
    getattr(lib(), "orc_softmax_blend_fwd_" + _sfx(dtype))(_p(col), _p(p2f), _p(ds), _p(zb), N, H, W, K,
                                                          _real(dtype, sigma), _real(dtype, gamma), _p(bg),
                                                          _real(dtype, znear), _real(dtype, zfar), _p(img))
    return img


def softmax_blend_bwd(colors, pix_to_face, dists, zbuf, grad_image, sigma, gamma, background, znear=1.0, zfar=100.0,
                      dtype=np.float32):
    col, p2f, ds, zb = _c(colors, dtype), _c(pix_to_face, np.int64), _c(dists, dtype), _c(zbuf, dtype)
    bg, gimg = _c(background, dtype), _c(grad_image, dtype)
    N, H, W, K = p2f.shape
    gcol, gds, gzb = np.empty_like(col), np.empty_like(ds), np.empty_like(zb)
    getattr(lib(), "orc_softmax_blend_bwd_" + _sfx(dtype))(_p(col), _p(p2f), _p(ds), _p(zb), _p(gimg), N, H, W, K,
                                                          _real(dtype, sigma), _real(dtype, gamma), _p(bg),
                                                          _real(dtype, znear), _real(dtype, zfar), _p(gcol), _p(gds),
                                                          _p(gzb))
    return gcol, gds, gzb


def mesh_regularizers(verts, edge_quads, nbr_offsets, nbr_index, dtype=np.float32):
    """-> per-mesh sums (normal, laplacian) and their gradients (N,V,3)"""
    v = _c(verts, dtype)
    q, off, idx = _c(edge_quads, np.int32), _c(nbr_offsets, np.int32), _c(nbr_index, np.int32)
    N, V = v.shape[:2]
    nsum, lsum = np.empty(N, dtype), np.empty(N, dtype)
    gN, gL = np.empty_like(v), np.empty_like(v)
    getattr(lib(), "orc_mesh_regularizers_" + _sfx(dtype))(_p(v), N, V, _p(q), q.shape[0], _p(off), _p(idx), _p(nsum),
                                                          _p(lsum), _p(gN), _p(gL))
    return nsum, lsum, gN, gL


def render_fwd_bwd(verts, faces, R, T, maps, verts_uvs, H, W, K, s, sigma=1e-4, background=(1., 1., 1.),
                   clip_inside=True, grad_image=None, n_threads=1, blur_radius=None, dtype=np.float32):
    """Whole path on the CPU.  verts (N,V,3), faces (F,3) shared topology, maps (N,Ht,Wt,3), verts_uvs (N,V,2).
    Returns image (N,4,H,W) and, when grad_image is given, (g_verts, g_R, g_T, g_maps)."""
    verts, R, T, maps, uvs = _c(verts, dtype), _c(R, dtype), _c(T, dtype), _c(maps, dtype), _c(verts_uvs, dtype)
    faces, bg = _c(faces, np.int64), _c(background, dtype)
    N, V = verts.shape[:2]
    F = faces.shape[0]
    Ht, Wt = maps.shape[1:3]
    blur = blur_radius_for(sigma) if blur_radius is None else blur_radius
    mode = blend_mode(sigma, clip_inside)
    img = np.empty((N, 4, H, W), dtype)
    gi = None if grad_image is None else _c(grad_image, dtype)
    gv, gR, gT, gm = np.zeros_like(verts), np.zeros_like(R), np.zeros_like(T), np.zeros_like(maps)
    getattr(lib(), "orc_render_fwd_bwd_" + _sfx(dtype))(
        _p(verts), _p(faces), _p(R), _p(T), _p(maps), _p(uvs), N, V, F, H, W, K, Ht, Wt, _real(dtype, s),
        _real(dtype, sigma), _real(dtype, blur), _p(bg), mode, int(blur > 0), None if gi is None else _p(gi), _p(img),
        _p(gv), _p(gR), _p(gT), _p(gm), n_threads)
    if gi is None:
        return img
    return img, (gv, gR, gT, gm)


def hardware_threads():
    return int(lib().orc_hardware_threads())
