"""Error behaviour of the C ABI (include/unicorn_render.h): every entry point validates its arguments BEFORE anything is
launched and reports through a negative code + ur_last_error(), the way the reference stack's native layer raises from
TORCH_CHECK (pytorch3d `_C.rasterize_meshes` & co., reached from renderer.py:48-49).  Nothing here launches a kernel
that reads the bad argument: a call that got past validation with a null pointer would fault the GPU."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

UR_ERR_INVALID_ARG, UR_ERR_UNSUPPORTED, UR_ERR_LAUNCH, UR_ERR_WORKSPACE = -1, -2, -3, -4

SIZE_HELPERS = {"ur_last_error", "ur_abi_version", "ur_workspace_bytes", "ur_layers_elems", "ur_backward_scratch_bytes",
                "ur_render_meshes_can_tile"}


@pytest.fixture(scope="module")
def native():
    from unicorn_amd import native as n
    n.lib()
    return n


def _zero(argtype):
    if argtype is ctypes.c_void_p:
        return None
    if argtype is ctypes.c_float:
        return 0.0
    return 0


def test_codes_match_the_header(native):
    import os
    import re
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "unicorn_render.h")).read()
    codes = {k: int(v) for k, v in re.findall(r"#define\s+(UR_ERR_[A-Z_]+)\s+\((-?\d+)\)", hdr)}
    assert codes == {"UR_ERR_INVALID_ARG": UR_ERR_INVALID_ARG, "UR_ERR_UNSUPPORTED": UR_ERR_UNSUPPORTED,
                     "UR_ERR_LAUNCH": UR_ERR_LAUNCH, "UR_ERR_WORKSPACE": UR_ERR_WORKSPACE}


def test_every_entry_point_rejects_null_pointers_and_zero_sizes(native):
    L = native.lib()
    torch.zeros(1, device="cuda")  # a current device
    checked = 0
    for name, (res, args) in native._SIGNATURES.items():
        if name in SIZE_HELPERS:
            continue
        rc = getattr(L, name)(*[_zero(a) for a in args])
        msg = L.ur_last_error().decode()
        assert rc < 0, f"{name} accepted null pointers / zero sizes (rc={rc})"
        assert msg, f"{name} failed without a message"
        checked += 1
    torch.cuda.synchronize()  # nothing faulted
    assert checked >= 18


def _raster_args(native, ws, ws_bytes, K=2, blur=0.0, perspective=0):
    N, F, H, W = 1, 1, 8, 8
    dev = "cuda"
    fv = torch.tensor([[[-0.5, -0.5, 1.0], [0.5, -0.5, 1.0], [0.0, 0.5, 1.0]]], device=dev)
    first = torch.zeros(N, dtype=torch.int64, device=dev)
    num = torch.ones(N, dtype=torch.int64, device=dev)
    p2f = torch.empty(N, H, W, K, dtype=torch.int32, device=dev)
    zb, di = torch.empty(N, H, W, K, device=dev), torch.empty(N, H, W, K, device=dev)
    ba = torch.empty(N, H, W, K, 3, device=dev)
    keep = (fv, first, num, p2f, zb, ba, di)
    stream = torch.cuda.current_stream().cuda_stream
    return keep, (fv.data_ptr(), first.data_ptr(), num.data_ptr(), N, F, H, W, blur, K, perspective, 0, 0, p2f.data_ptr(),
                  zb.data_ptr(), ba.data_ptr(), di.data_ptr(), ws, ws_bytes, stream)


def test_workspace_errors(native):
    L = native.lib()
    need = L.ur_workspace_bytes(1, 1, 8, 8, 2)
    assert need > 0
    buf = torch.empty(need + 64, dtype=torch.uint8, device="cuda")
    base = buf.data_ptr()
    assert base % 16 == 0
    keep, args = _raster_args(native, base, need - 1)
    assert L.ur_rasterize_meshes(*args) == UR_ERR_WORKSPACE and b"too small" in L.ur_last_error()
    keep, args = _raster_args(native, base + 4, need)
    assert L.ur_rasterize_meshes(*args) == UR_ERR_WORKSPACE and b"aligned" in L.ur_last_error()
    keep, args = _raster_args(native, None, need)
    assert L.ur_rasterize_meshes(*args) == UR_ERR_WORKSPACE
    keep, args = _raster_args(native, base, need)
    assert L.ur_rasterize_meshes(*args) == 0        # and the same call with a good workspace goes through
    torch.cuda.synchronize()
    assert int((keep[3] >= 0).sum()) > 0             # the triangle was rasterized


def test_argument_errors(native):
    L = native.lib()
    need = L.ur_workspace_bytes(1, 1, 8, 8, 2)
    buf = torch.empty(need, dtype=torch.uint8, device="cuda")
    keep, args = _raster_args(native, buf.data_ptr(), need, blur=-1.0)
    assert L.ur_rasterize_meshes(*args) == UR_ERR_INVALID_ARG and b"blur_radius" in L.ur_last_error()
    keep, args = _raster_args(native, buf.data_ptr(), need, blur=float("nan"))
    assert L.ur_rasterize_meshes(*args) == UR_ERR_INVALID_ARG
    keep, args = _raster_args(native, buf.data_ptr(), need, perspective=1)
    assert L.ur_rasterize_meshes(*args) == UR_ERR_UNSUPPORTED and b"perspective_correct" in L.ur_last_error()
    # the blend of caller-computed colours: kinds 0..3, soft kinds need sigma > 0, softmax needs its depth arguments
    N, H, W, K = 1, 4, 4, 2
    col = torch.zeros(N, H, W, K, 3, device="cuda")
    p2f = torch.zeros(N, H, W, K, dtype=torch.int32, device="cuda")
    d = torch.zeros(N, H, W, K, device="cuda")
    img = torch.empty(N, 4, H, W, device="cuda")
    bg = (ctypes.c_float * 3)(1, 1, 1)
    st = torch.cuda.current_stream().cuda_stream

    def blend(kind, sigma, zbuf=None, gamma=1e-4, znear=1.0, zfar=100.0):
        return L.ur_blend_colors(col.data_ptr(), p2f.data_ptr(), d.data_ptr(), zbuf, N, H, W, K, kind, sigma, gamma,
                                 ctypes.cast(bg, ctypes.c_void_p), znear, zfar, img.data_ptr(), st)
    assert blend(7, 1e-4) == UR_ERR_INVALID_ARG and b"blend kind" in L.ur_last_error()
    assert blend(2, 0.0) == UR_ERR_INVALID_ARG and b"sigma" in L.ur_last_error()
    assert blend(3, 1e-4) == UR_ERR_INVALID_ARG and b"zbuf" in L.ur_last_error()
    assert blend(3, 1e-4, zbuf=d.data_ptr(), gamma=0.0) == UR_ERR_INVALID_ARG
    assert blend(3, 1e-4, zbuf=d.data_ptr(), znear=2.0, zfar=1.0) == UR_ERR_INVALID_ARG
    assert blend(2, 1e-4) == 0
    torch.cuda.synchronize()
