"""ctypes binding of the C ABI in include/unicorn_render.h (built in-tree as unicorn_amd/lib/libunicorn_render.so).

There is NO fallback: if the shared library is missing or a call fails, this module raises.  Tensors are handed over
as raw device pointers; kernels are enqueued on torch's current HIP stream of the tensors' device.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# UNICORN_RENDER_LIB: load another build of the same library (A/B timing of two builds on one box)
LIB_PATH = os.environ.get("UNICORN_RENDER_LIB") or os.path.join(_HERE, "lib", "libunicorn_render.so")

BLEND_HARD, BLEND_CLIP_INSIDE, BLEND_SIGMOID, BLEND_SOFTMAX = 0, 1, 2, 3

_lib = None

_vp, _i, _i64, _f, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t
_F3 = ctypes.c_float * 3

_SIGNATURES = {
    "ur_last_error": (ctypes.c_char_p, []),
    "ur_abi_version": (_i, []),
    "ur_profile_begin": (_i, [_i]),
    "ur_profile_end": (_i, [_vp, _vp, _i, _vp]),
    "ur_workspace_bytes": (_sz, [_i64, _i, _i, _i, _i]),
    "ur_project_faces": (_i, [_vp, _vp, _i64, _vp, _vp, _f, _f, _i, _i, _i, _vp, _vp]),
    "ur_project_faces_backward": (_i, [_vp, _vp, _i64, _vp, _vp, _f, _f, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "ur_rasterize_meshes": (_i, [_vp, _vp, _vp, _i, _i64, _i, _i, _f, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz,
                                 _vp]),
    "ur_rasterize_meshes_static": (_i, [_vp, _vp, _vp, _i, _i64, _i, _i, _f, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "ur_rasterize_meshes_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _i, _i, _vp, _vp]),
    "ur_tile_faces": (_i, [_vp, _vp, _vp, _i, _i64, _i, _i, _f, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "ur_shade_blend": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _i, _vp, _vp]),
    "ur_blend_colors": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _vp, _f, _f, _vp, _vp]),
    "ur_blend_colors_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _vp, _f, _f, _vp, _vp, _vp,
                                      _vp]),
    "ur_mesh_regularizers": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ur_layers_elems": (_sz, [_i, _i, _i, _i]),
    "ur_selftest_division": (_i, [ctypes.c_uint64, ctypes.c_uint32, _vp, _vp]),
    "ur_render_layered_forward": (_i, [_vp, _vp, _vp, _i, _i64, _i, _i, _f, _i, _i, _i, _vp, _vp, _i, _i, _i, _f, _vp,
                                       _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "ur_backward_scratch_bytes": (_sz, [_i, _i, _i, _i]),
    "ur_render_layered_backward": (_i, [_vp, _vp, _vp, _i, _i64, _i, _i, _f, _i, _i, _vp, _vp, _i, _i, _i, _f, _vp,
                                        _vp, _i, _vp, _vp, _sz, _vp, _sz, _vp, _vp, _vp]),
    "ur_render_meshes_can_tile": (_i, [_i, _i, _i, _i]),
    "ur_blend_intermediates": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp]),
    "ur_render_meshes_forward": (_i, [_vp, _vp, _i64, _vp, _vp, _f, _f, _i, _i, _i, _i, _vp, _vp, _i, _i, _f, _i, _i,
                                      _i, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                                      _vp]),
    "ur_render_meshes_backward": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _f, _f, _i, _i, _i, _i, _vp, _vp, _i, _i, _f, _i,
                                       _i, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _i, _vp, _vp, _sz, _vp, _sz, _vp, _vp,
                                       _i, _vp, _vp, _vp, _vp]),
    "ur_shade_blend_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _i,
                                     _vp, _vp, _vp, _vp]),
}


def source_hash():
    """sha256 over the kernel sources the library is built from (ur_render.hip + ur_device.h): stamps counter files,
    so that a counter taken on other kernels is not mistaken for this build's"""
    import hashlib
    h = hashlib.sha256()
    for name in ("ur_render.hip", "ur_device.h"):
        with open(os.path.join(_HERE, "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def exported_symbols():
    """Names include/unicorn_render.h declares (checked against the built library by the CPU tests)."""
    return sorted(_SIGNATURES)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                               f"g.build()'` (hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {lib().ur_last_error().decode()}")


# ---- optional per-call device timing (bench.py's roofline leg): HIP events recorded on the launch stream around a
# C-ABI call; nothing is recorded unless enabled, and nothing here synchronises.
_timing = None


def enable_timing(on=True):
    global _timing
    _timing = {} if on else None


class _timed:
    def __init__(self, name, t):
        self.name, self.t = name, t

    def __enter__(self):
        if _timing is not None:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record(torch.cuda.current_stream(self.t.device))

    def __exit__(self, *a):
        if _timing is not None:
            self.e1.record(torch.cuda.current_stream(self.t.device))
            _timing.setdefault(self.name, []).append((self.e0, self.e1))


def timing_summary():
    """{call name: (launches, mean ms)}; synchronises."""
    torch.cuda.synchronize()
    return {k: (len(v), sum(a.elapsed_time(b) for a, b in v) / len(v)) for k, v in (_timing or {}).items()}


def _ptr(t, dtype=None, allow_none=False):
    if t is None:
        if allow_none:
            return None
        raise ValueError("missing tensor")
    if t.is_cuda and t.is_contiguous() and (dtype is None or t.dtype is dtype):  # the usual case, one test
        return t.data_ptr()
    if not t.is_cuda:
        raise ValueError("unicorn_amd kernels need tensors on the GPU (got a CPU tensor); there is no CPU path")
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    raise ValueError(f"expected {dtype}, got {t.dtype}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(t):
    """torch's current HIP stream on t's device, as the raw handle the C ABI takes"""
    if _raw_stream is not None:
        return _raw_stream(t.device.index)
    return torch.cuda.current_stream(t.device).cuda_stream


class _on:
    """`with _on(t):` = make t's GPU current, after checking that t lives on one.  When it is current already (the
    usual case: one process per GPU) nothing is switched."""

    def __init__(self, t):
        _ptr(t)
        self._ctx = None if t.device.index == torch.cuda.current_device() else torch.cuda.device(t.device)

    def __enter__(self):
        if self._ctx is not None:
            self._ctx.__enter__()

    def __exit__(self, *a):
        if self._ctx is not None:
            self._ctx.__exit__(*a)


_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _call(name, t, *args):
    """one C-ABI call on t's GPU.  The usual case -- t's GPU is current, no timing asked for -- goes straight to the
    library; otherwise through the device switch / the event pair."""
    fn = getattr(lib(), name)
    if _timing is None and _cur_device is not None and t.is_cuda and t.device.index == _cur_device():
        rc = fn(*args)
    else:
        with _on(t), _timed(name, t):
            rc = fn(*args)
    if rc != 0:
        _check(rc, name)


_size_cache = {}


def _sized(fn_name, *args):
    """a size query of the library (pure functions of their integer arguments), asked once per argument tuple"""
    key = (fn_name,) + args
    v = _size_cache.get(key)
    if v is None:
        v = _size_cache[key] = int(getattr(lib(), fn_name)(*args))
    return v


def workspace_bytes(total_faces, N, H, W, K):
    return _sized("ur_workspace_bytes", int(total_faces), int(N), int(H), int(W), int(K))


def _workspace(total_faces, device, N, H, W, K):
    return torch.empty(workspace_bytes(total_faces, N, H, W, K), dtype=torch.uint8, device=device)


def _faces_layout(faces, N):
    if faces.dim() == 2:
        return int(faces.shape[0]), 0
    if faces.dim() == 3 and faces.shape[0] == N:
        return int(faces.shape[1]), int(faces.shape[1]) * 3
    raise ValueError("faces must be (F,3) or (N,F,3)")


def project_faces(verts, faces, R, T, sx, sy):
    """verts (N,V,3) f32, faces i64 (F,3) shared or (N,F,3) per mesh, R (N,3,3), T (N,3) -> face_verts (N*F,3,3)"""
    N, V = verts.shape[:2]
    F, fstride = _faces_layout(faces, N)
    out = torch.empty(N * F, 3, 3, dtype=torch.float32, device=verts.device)
    with _on(verts), _timed("ur_project_faces", verts):
        _check(lib().ur_project_faces(_ptr(verts, torch.float32), _ptr(faces, torch.int64), fstride,
                                      _ptr(R, torch.float32), _ptr(T, torch.float32), sx, sy, N, V, F, _ptr(out),
                                      _stream(verts)), "ur_project_faces")
    return out


def project_faces_backward(verts, faces, R, T, sx, sy, grad_face_verts, need_verts=True, need_R=True, need_T=True):
    N, V = verts.shape[:2]
    F, fstride = _faces_layout(faces, N)
    gv = torch.empty_like(verts) if need_verts else None
    gR = torch.empty_like(R) if need_R else None
    gT = torch.empty_like(T) if need_T else None
    with _on(verts), _timed("ur_project_faces_backward", verts):
        _check(lib().ur_project_faces_backward(_ptr(verts, torch.float32), _ptr(faces, torch.int64), fstride,
                                               _ptr(R, torch.float32), _ptr(T, torch.float32), sx, sy, N, V, F,
                                               _ptr(grad_face_verts, torch.float32), _ptr(gv, None, True),
                                               _ptr(gR, None, True), _ptr(gT, None, True), _stream(verts)),
               "ur_project_faces_backward")
    return gv, gR, gT


def rasterize_meshes(face_verts, mesh_first_face, mesh_num_faces, image_size, blur_radius, faces_per_pixel,
                     clip_barycentric_coords, cull_backfaces=False, perspective_correct=False, want_zbuf=True,
                     want_bary=True, want_dists=True):
    """face_verts (NF,3,3) -> pix_to_face i32 (N,H,W,K), zbuf, bary (N,H,W,K,3), dists"""
    _ptr(face_verts, torch.float32)
    H, W = image_size
    N, K = int(mesh_first_face.shape[0]), int(faces_per_pixel)
    NF = int(face_verts.shape[0])
    dev = face_verts.device
    p2f = torch.empty(N, H, W, K, dtype=torch.int32, device=dev)
    zbuf = torch.empty(N, H, W, K, dtype=torch.float32, device=dev) if want_zbuf else None
    bary = torch.empty(N, H, W, K, 3, dtype=torch.float32, device=dev) if want_bary else None
    dists = torch.empty(N, H, W, K, dtype=torch.float32, device=dev) if want_dists else None
    ws = _workspace(NF, dev, N, H, W, K)
    with _on(face_verts), _timed("ur_rasterize_meshes", face_verts):
        _check(lib().ur_rasterize_meshes(_ptr(face_verts, torch.float32), _ptr(mesh_first_face, torch.int64),
                                         _ptr(mesh_num_faces, torch.int64), N, NF, H, W, float(blur_radius), K,
                                         int(perspective_correct), int(clip_barycentric_coords), int(cull_backfaces),
                                         _ptr(p2f), _ptr(zbuf, None, True), _ptr(bary, None, True),
                                         _ptr(dists, None, True), _ptr(ws), ws.numel(), _stream(face_verts)),
               "ur_rasterize_meshes")
    return p2f, zbuf, bary, dists


STATIC_SCRATCH_BYTES = 64 << 20  # UR_STATIC_SCRATCH_BYTES


def rasterize_meshes_static(face_verts, mesh_first_face, mesh_num_faces, image_size, blur_radius, faces_per_pixel,
                            clip_barycentric_coords, out, cull_backfaces=False, perspective_correct=False):
    """ur_rasterize_meshes_static: the allocation-free variant.  `out` = (pix_to_face i32, zbuf, bary, dists), tensors
    the CALLER holds ((N,H,W,K[,3]); any of the last three may be None); nothing is allocated here or in the library."""
    H, W = image_size
    N, K, NF = int(mesh_first_face.shape[0]), int(faces_per_pixel), int(face_verts.shape[0])
    p2f, zbuf, bary, dists = out
    for t, shape in ((p2f, (N, H, W, K)), (zbuf, (N, H, W, K)), (bary, (N, H, W, K, 3)), (dists, (N, H, W, K))):
        if t is not None and tuple(t.shape) != shape:
            raise ValueError(f"output tensor has shape {tuple(t.shape)}, expected {shape}")
    with _on(face_verts):
        _check(lib().ur_rasterize_meshes_static(_ptr(face_verts, torch.float32), _ptr(mesh_first_face, torch.int64),
                                                _ptr(mesh_num_faces, torch.int64), N, NF, H, W, float(blur_radius), K,
                                                int(perspective_correct), int(clip_barycentric_coords),
                                                int(cull_backfaces), _ptr(p2f, torch.int32),
                                                _ptr(zbuf, torch.float32, True), _ptr(bary, torch.float32, True),
                                                _ptr(dists, torch.float32, True), _stream(face_verts)),
               "ur_rasterize_meshes_static")
    return out


def rasterize_meshes_backward(face_verts, pix_to_face, grad_zbuf, grad_bary, grad_dists, clip_barycentric_coords,
                              perspective_correct=False):
    N, H, W, K = pix_to_face.shape
    gfv = torch.zeros_like(face_verts)
    with _on(face_verts), _timed("ur_rasterize_meshes_backward", face_verts):
        _check(lib().ur_rasterize_meshes_backward(_ptr(face_verts, torch.float32), _ptr(pix_to_face, torch.int32),
                                                  _ptr(grad_zbuf, torch.float32, True),
                                                  _ptr(grad_bary, torch.float32, True),
                                                  _ptr(grad_dists, torch.float32, True), N, face_verts.shape[0], H, W,
                                                  K, int(perspective_correct), int(clip_barycentric_coords),
                                                  _ptr(gfv), _stream(face_verts)),
               "ur_rasterize_meshes_backward")
    return gfv


def tile_faces(face_verts, mesh_first_face, mesh_num_faces, image_size, blur_radius, cap, cull_backfaces=False):
    H, W = image_size
    N, NF = int(mesh_first_face.shape[0]), int(face_verts.shape[0])
    dev = face_verts.device
    TH, TW = (H + 7) // 8, (W + 7) // 8
    counts = torch.empty(N, TH, TW, dtype=torch.int32, device=dev)
    lists = torch.empty(N, TH, TW, cap, dtype=torch.int32, device=dev)
    ws = _workspace(NF, dev, N, H, W, 1)
    with _on(face_verts), _timed("ur_tile_faces", face_verts):
        _check(lib().ur_tile_faces(_ptr(face_verts, torch.float32), _ptr(mesh_first_face, torch.int64),
                                   _ptr(mesh_num_faces, torch.int64), N, NF, H, W, float(blur_radius),
                                   int(cull_backfaces), int(cap), _ptr(counts), _ptr(lists), _ptr(ws), ws.numel(),
                                   _stream(face_verts)), "ur_tile_faces")
    return counts, lists


_rgb_cache = {}


def _rgb(v):
    """three floats as a C array (kept per distinct colour: the arrays are read-only inputs of the calls)"""
    if v is None:
        return None
    key = tuple(v) if not isinstance(v, tuple) else v
    a = _rgb_cache.get(key)
    if a is None:
        if len(_rgb_cache) > 256:
            _rgb_cache.pop(next(iter(_rgb_cache)))  # one entry out (the oldest); callers hold their arrays anyway
        a = _rgb_cache[key] = _F3(*[float(x) for x in key])
    return a


def shade_blend(pix_to_face, bary, dists, maps, face_uvs, face_uvs_shared, mesh_first_face, sigma, background,
                ambient, mode):
    N, H, W, K = pix_to_face.shape
    Ht, Wt = maps.shape[1:3]
    image = torch.empty(N, 4, H, W, dtype=torch.float32, device=maps.device)
    with _on(maps), _timed("ur_shade_blend", maps):
        _check(lib().ur_shade_blend(_ptr(pix_to_face, torch.int32), _ptr(bary, torch.float32),
                                    _ptr(dists, torch.float32), _ptr(maps, torch.float32),
                                    _ptr(face_uvs, torch.float32), int(face_uvs_shared),
                                    _ptr(mesh_first_face, torch.int64, True), N, H, W, K, Ht, Wt, float(sigma),
                                    ctypes.cast(_rgb(background), _vp), ctypes.cast(_rgb(ambient), _vp)
                                    if ambient is not None else None, int(mode), _ptr(image), _stream(maps)),
               "ur_shade_blend")
    return image


def shade_blend_backward(grad_image, pix_to_face, bary, dists, maps, face_uvs, face_uvs_shared, mesh_first_face,
                         sigma, background, ambient, mode, need_maps=True):
    N, H, W, K = pix_to_face.shape
    Ht, Wt = maps.shape[1:3]
    gmaps = torch.zeros_like(maps) if need_maps else None
    gbary = torch.empty_like(bary)
    gdists = torch.empty_like(dists)
    with _on(maps), _timed("ur_shade_blend_backward", maps):
        _check(lib().ur_shade_blend_backward(_ptr(grad_image, torch.float32), _ptr(pix_to_face, torch.int32),
                                             _ptr(bary, torch.float32), _ptr(dists, torch.float32),
                                             _ptr(maps, torch.float32), _ptr(face_uvs, torch.float32),
                                             int(face_uvs_shared), _ptr(mesh_first_face, torch.int64, True), N, H, W,
                                             K, Ht, Wt, float(sigma), ctypes.cast(_rgb(background), _vp),
                                             ctypes.cast(_rgb(ambient), _vp) if ambient is not None else None,
                                             int(mode), _ptr(gmaps, None, True), _ptr(gbary), _ptr(gdists),
                                             _stream(maps)), "ur_shade_blend_backward")
    return gmaps, gbary, gdists


def blend_colors(colors, pix_to_face, dists, zbuf, kind, sigma, gamma, background, znear=1.0, zfar=100.0):
    """per-fragment colours (N,H,W,K,3) -> image (N,4,H,W); kind BLEND_HARD / CLIP_INSIDE / SIGMOID / SOFTMAX"""
    N, H, W, K = pix_to_face.shape
    image = torch.empty(N, 4, H, W, dtype=torch.float32, device=colors.device)
    with _on(colors), _timed("ur_blend_colors", colors):
        _check(lib().ur_blend_colors(_ptr(colors, torch.float32), _ptr(pix_to_face, torch.int32),
                                     _ptr(dists, torch.float32), _ptr(zbuf, torch.float32, True), N, H, W, K,
                                     int(kind), float(sigma), float(gamma), ctypes.cast(_rgb(background), _vp),
                                     float(znear), float(zfar), _ptr(image), _stream(colors)), "ur_blend_colors")
    return image


def blend_colors_backward(grad_image, colors, pix_to_face, dists, zbuf, kind, sigma, gamma, background, znear=1.0,
                          zfar=100.0):
    N, H, W, K = pix_to_face.shape
    gcol, gd = torch.empty_like(colors), torch.empty_like(dists)
    gz = torch.empty_like(dists) if zbuf is not None else None
    with _on(colors), _timed("ur_blend_colors_backward", colors):
        _check(lib().ur_blend_colors_backward(_ptr(grad_image, torch.float32), _ptr(colors, torch.float32),
                                              _ptr(pix_to_face, torch.int32), _ptr(dists, torch.float32),
                                              _ptr(zbuf, torch.float32, True), N, H, W, K, int(kind), float(sigma),
                                              float(gamma), ctypes.cast(_rgb(background), _vp), float(znear),
                                              float(zfar), _ptr(gcol), _ptr(gd), _ptr(gz, None, True),
                                              _stream(colors)), "ur_blend_colors_backward")
    return gcol, gd, gz


def render_layered_forward(face_verts, mesh_first_face, mesh_num_faces, image_size, blur_radius, faces_per_pixel,
                           clip_barycentric_coords, maps, face_uvs, face_uvs_shared, sigma, background, ambient, mode,
                           cull_backfaces=False):
    """-> image (N,4,H,W), layers (int32, opaque), workspace (uint8, opaque); the last two feed the backward"""
    _ptr(face_verts, torch.float32)
    H, W = image_size
    N, K, NF = int(mesh_first_face.shape[0]), int(faces_per_pixel), int(face_verts.shape[0])
    Ht, Wt = maps.shape[1:3]
    dev = face_verts.device
    image = torch.empty(N, 4, H, W, dtype=torch.float32, device=dev)
    layers = torch.empty(_sized("ur_layers_elems", N, H, W, K), dtype=torch.int32, device=dev)
    ws = _workspace(NF, dev, N, H, W, K)
    with _on(face_verts), _timed("ur_render_layered_forward", face_verts):
        _check(lib().ur_render_layered_forward(
            _ptr(face_verts, torch.float32), _ptr(mesh_first_face, torch.int64), _ptr(mesh_num_faces, torch.int64), N,
            NF, H, W, float(blur_radius), K, int(clip_barycentric_coords), int(cull_backfaces),
            _ptr(maps, torch.float32), _ptr(face_uvs, torch.float32), int(face_uvs_shared), Ht, Wt, float(sigma),
            ctypes.cast(_rgb(background), _vp), ctypes.cast(_rgb(ambient), _vp) if ambient is not None else None,
            int(mode), _ptr(image), _ptr(layers), _ptr(ws), ws.numel(), _stream(face_verts)),
            "ur_render_layered_forward")
    return image, layers, ws


def render_layered_backward(grad_image, mesh_first_face, mesh_num_faces, total_faces, image_size, blur_radius,
                            faces_per_pixel, clip_barycentric_coords, maps, face_uvs, face_uvs_shared, sigma,
                            background, ambient, mode, layers, workspace, need_face_verts=True, need_maps=True):
    H, W = image_size
    N, K = int(mesh_first_face.shape[0]), int(faces_per_pixel)
    Ht, Wt = maps.shape[1:3]
    dev = maps.device
    # both accumulators come out of ONE zero-filled allocation (one fill launch instead of two)
    n_fv = total_faces * 9 if need_face_verts else 0
    n_fv_pad = (n_fv + 63) // 64 * 64
    zeros = torch.zeros(n_fv_pad + (maps.numel() if need_maps else 0), dtype=torch.float32, device=dev)
    gfv = zeros[:n_fv].view(total_faces, 3, 3) if need_face_verts else None
    gmaps = zeros[n_fv_pad:].view(maps.shape) if need_maps else None
    scratch = torch.empty(_sized("ur_backward_scratch_bytes", N, H, W, K), dtype=torch.uint8, device=dev)
    with _on(maps), _timed("ur_render_layered_backward", maps):
        _check(lib().ur_render_layered_backward(
            _ptr(grad_image, torch.float32), _ptr(mesh_first_face, torch.int64), _ptr(mesh_num_faces, torch.int64), N,
            int(total_faces), H, W, float(blur_radius), K, int(clip_barycentric_coords), _ptr(maps, torch.float32),
            _ptr(face_uvs, torch.float32), int(face_uvs_shared), Ht, Wt, float(sigma),
            ctypes.cast(_rgb(background), _vp), ctypes.cast(_rgb(ambient), _vp) if ambient is not None else None,
            int(mode), _ptr(layers, torch.int32), _ptr(workspace), workspace.numel(), _ptr(scratch), scratch.numel(),
            _ptr(gfv, None, True),
            _ptr(gmaps, None, True), _stream(maps)), "ur_render_layered_backward")
    return gfv, gmaps


def _rgb_ptr(arr):
    """address of a colour array from _rgb(); the CALLER keeps `arr` in a local for the duration of the C call (the
    cache above may evict it meanwhile)"""
    return None if arr is None else ctypes.addressof(arr)


_step_layouts = {}


def _step_layout(N, F, H, W, K):
    """byte offsets inside the one allocation a render step keeps between its two calls:
    [face_verts | layers | workspace | backward scratch | face-gradient accumulator], each part 256-byte aligned
    -> (layers, workspace, workspace bytes, scratch, scratch bytes, accumulator, total)"""
    key = (N, F, H, W, K)
    v = _step_layouts.get(key)
    if v is None:
        al = lambda x: (x + 255) // 256 * 256
        NF = N * F
        o_lay = al(NF * 36)
        o_ws = o_lay + al(_sized("ur_layers_elems", N, H, W, K) * 4)
        ws_b = workspace_bytes(NF, N, H, W, K)
        o_scr = o_ws + al(ws_b)
        scr_b = _sized("ur_backward_scratch_bytes", N, H, W, K)
        o_acc = o_scr + al(scr_b)
        v = _step_layouts[key] = (o_lay, o_ws, ws_b, o_scr, scr_b, o_acc, o_acc + al(NF * 36))
    return v


def step_face_verts(state, N, F):
    """the projected corners (N*F,3,3) ur_render_meshes_forward left in a step's allocation"""
    return state[:N * F * 36].view(torch.float32).view(N * F, 3, 3)


def render_meshes_forward(verts, faces, R, T, sx, sy, mesh_first_face, mesh_num_faces, image_size, blur_radius,
                          faces_per_pixel, clip_barycentric_coords, maps, face_uvs, face_uvs_shared, sigma, background,
                          ambient, mode, need_geometry_grads=False, need_map_grads=False, cull_backfaces=False):
    """equal-sized meshes: camera + rasterizer + shader in one call.  -> image (N,4,H,W), state, grad_maps.
    `state`: ONE uint8 allocation holding everything the step keeps for its backward (_step_layout: projected corners,
    layers, workspace, the backward's scratch and its face-gradient accumulator) -- the host side of a step is a few
    allocator calls and two C calls, so each allocation saved shows in the eager rate.  `grad_maps`: the texture
    gradient's buffer, cleared on the way (None unless need_map_grads); the accumulator is cleared likewise when
    need_geometry_grads."""
    M, V = verts.shape[:2]   # M meshes shown under N = R.shape[0] poses (image n = mesh n % M; M == N: one each)
    N = int(R.shape[0])
    F, fstride = _faces_layout(faces, M)
    H, W = image_size
    K = int(faces_per_pixel)
    Ht, Wt = maps.shape[1:3]
    dev = verts.device
    o_lay, o_ws, ws_b, _, _, o_acc, total = _step_layout(N, F, H, W, K)
    image = torch.empty(N, 4, H, W, dtype=torch.float32, device=dev)
    state = torch.empty(total, dtype=torch.uint8, device=dev)
    gmaps = torch.empty_like(maps) if need_map_grads else None
    base = state.data_ptr()
    bg_arr, amb_arr = _rgb(background), _rgb(ambient)  # held here until the call has returned
    _call("ur_render_meshes_forward", verts,
            _ptr(verts, torch.float32), _ptr(faces, torch.int64), fstride, _ptr(R, torch.float32),
            _ptr(T, torch.float32), sx, sy, N, M, V, F, _ptr(mesh_first_face, torch.int64),
            _ptr(mesh_num_faces, torch.int64), H, W, float(blur_radius), K, int(clip_barycentric_coords),
            int(cull_backfaces), _ptr(maps, torch.float32), _ptr(face_uvs, torch.float32), int(face_uvs_shared), Ht, Wt,
            float(sigma), _rgb_ptr(bg_arr), _rgb_ptr(amb_arr), int(mode), base, image.data_ptr(), base + o_lay,
            base + o_acc if need_geometry_grads else None, None if gmaps is None else gmaps.data_ptr(), base + o_ws,
            ws_b, _stream(verts))
    return image, state, gmaps


def render_meshes_backward(grad_image, verts, faces, R, T, sx, sy, mesh_first_face, mesh_num_faces, image_size,
                           blur_radius, faces_per_pixel, clip_barycentric_coords, maps, face_uvs, face_uvs_shared, sigma,
                           background, ambient, mode, state, grad_maps, grads_are_zero, need_verts, need_R, need_T,
                           need_maps):
    """-> grad_verts, grad_R, grad_T, grad_maps (None where not needed).  `state` / `grad_maps`: what the forward
    returned; grads_are_zero: they still are as the forward left them (False on a second backward through the same
    step: the call then clears the accumulators itself, and grad_maps may be None = allocate one)."""
    M, V = verts.shape[:2]
    N = int(R.shape[0])
    F, fstride = _faces_layout(faces, M)
    H, W = image_size
    K = int(faces_per_pixel)
    Ht, Wt = maps.shape[1:3]
    o_lay, o_ws, ws_b, o_scr, scr_b, o_acc, total = _step_layout(N, F, H, W, K)
    assert state.numel() == total
    geometry = need_verts or need_R or need_T
    if need_maps and grad_maps is None:
        grad_maps, grads_are_zero = torch.empty_like(maps), False
    gv = torch.empty_like(verts) if need_verts else None
    gR = torch.empty_like(R) if need_R else None
    gT = torch.empty_like(T) if need_T else None
    base = state.data_ptr()
    bg_arr, amb_arr = _rgb(background), _rgb(ambient)  # held here until the call has returned
    _call("ur_render_meshes_backward", verts,
            _ptr(grad_image, torch.float32), _ptr(verts, torch.float32), _ptr(faces, torch.int64), fstride,
            _ptr(R, torch.float32), _ptr(T, torch.float32), sx, sy, N, M, V, F, _ptr(mesh_first_face, torch.int64),
            _ptr(mesh_num_faces, torch.int64), H, W, float(blur_radius), K, int(clip_barycentric_coords),
            _ptr(maps, torch.float32), _ptr(face_uvs, torch.float32), int(face_uvs_shared), Ht, Wt, float(sigma),
            _rgb_ptr(bg_arr), _rgb_ptr(amb_arr), int(mode), base + o_lay, base + o_ws, ws_b, base + o_scr, scr_b,
            base + o_acc if geometry else None, grad_maps.data_ptr() if need_maps else None, int(bool(grads_are_zero)),
            None if gv is None else gv.data_ptr(), None if gR is None else gR.data_ptr(),
            None if gT is None else gT.data_ptr(), _stream(verts))
    return gv, gR, gT, (grad_maps if need_maps else None)


def can_tile(V, F, H, W):
    """may B meshes be shown under K*B poses without copying them (ur_render_meshes_* with n_meshes < N)?"""
    return bool(_sized("ur_render_meshes_can_tile", int(V), int(F), int(H), int(W)))


def blend_intermediates(pix_to_face, dists, sigma, mode):
    """-> alpha, occ_alpha (N,H,W,K+1): the layered blend's per-layer intermediates (LayeredShader debug=True)"""
    N, H, W, K = pix_to_face.shape
    alpha = torch.empty(N, H, W, K + 1, dtype=torch.float32, device=dists.device)
    occ = torch.empty_like(alpha)
    with _on(dists):
        _check(lib().ur_blend_intermediates(_ptr(pix_to_face, torch.int32), _ptr(dists, torch.float32), N, H, W, K,
                                            float(sigma), int(mode), _ptr(alpha), _ptr(occ), _stream(dists)),
               "ur_blend_intermediates")
    return alpha, occ


def mesh_regularizers(verts, edge_quads, nbr_offsets, nbr_index, need_grad=True):
    """verts (N,V,3) -> per-mesh sums (normal, laplacian) and, when asked, their gradients (N,V,3) each"""
    N, V = verts.shape[:2]
    dev = verts.device
    nsum = torch.empty(N, dtype=torch.float32, device=dev)
    lsum = torch.empty(N, dtype=torch.float32, device=dev)
    gN = torch.empty_like(verts) if need_grad else None
    gL = torch.empty_like(verts) if need_grad else None
    scratch = torch.empty_like(verts)
    with _on(verts), _timed("ur_mesh_regularizers", verts):
        _check(lib().ur_mesh_regularizers(_ptr(verts, torch.float32), N, V, _ptr(edge_quads, torch.int32),
                                          int(edge_quads.shape[0]), _ptr(nbr_offsets, torch.int32),
                                          _ptr(nbr_index, torch.int32), _ptr(nsum), _ptr(lsum), _ptr(gN, None, True),
                                          _ptr(gL, None, True), _ptr(scratch), _stream(verts)),
               "ur_mesh_regularizers")
    return nsum, lsum, gN, gL


def profile_begin(max_launches):
    """start bracketing the fused forward / backward kernels with HIP events on their launch stream"""
    _check(lib().ur_profile_begin(int(max_launches)), "ur_profile_begin")


def profile_end(max_launches):
    """-> {"k_raster<2>": [ms, ...], "k_render_bwd": [ms, ...]} for the launches since profile_begin"""
    ms = (ctypes.c_float * max_launches)()
    kinds = (ctypes.c_int * max_launches)()
    count = ctypes.c_int(0)
    _check(lib().ur_profile_end(ms, kinds, int(max_launches), ctypes.byref(count)), "ur_profile_end")
    names = {1: "k_raster<2>", 2: "k_render_bwd"}
    out = {}
    for i in range(count.value):
        out.setdefault(names.get(kinds[i], "other"), []).append(float(ms[i]))
    return out


def selftest_division(n, seed=0, device="cuda"):
    """number of (a, b) pairs, out of n pseudo-random ones, where the kernels' exact-division sequence differs from
    the hardware's IEEE division (expected 0)"""
    out = torch.zeros(1, dtype=torch.int64, device=device)
    with _on(out):
        _check(lib().ur_selftest_division(int(n), int(seed), _ptr(out), _stream(out)), "ur_selftest_division")
    return int(out.item())
