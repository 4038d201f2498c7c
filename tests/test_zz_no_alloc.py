"""The allocation-free entry point ur_rasterize_meshes_static (no counterpart in the reference; for callers that cannot
allocate: HIP graph capture, serving loops) against the entry point it is a variant of, ur_rasterize_meshes: same
fragments bit for bit, on two calls in a row, the second with a smaller input (the static scratch keeps the first
call's contents), and a clear error beyond the scratch's size."""
import numpy as np
import pytest
import torch

from oracle import orc
from unicorn_amd import synth

pytestmark = pytest.mark.gpu

S30 = float(1.0 / np.tan(np.deg2rad(15.0)))


def _inputs(N, level, seed):
    sc = synth.make_scene(N, level=level, seed=seed)
    vn = orc.project_verts(sc["verts"].numpy(), sc["R"].numpy(), sc["T"].numpy(), S30)
    faces = sc["faces"].numpy()
    F = faces.shape[0]
    fv = np.stack([vn[n][faces] for n in range(N)]).reshape(-1, 3, 3).astype(np.float32)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return fv, dev(fv), dev(np.arange(N, dtype=np.int64) * F), dev(np.full(N, F, dtype=np.int64))


def test_static_scratch_rasterizer_equals_the_workspace_one():
    from unicorn_amd import native
    blur = orc.blur_radius_for(1e-4)
    # first a larger call (K=50: the K-list spill area inside the scratch is in use), then a smaller one
    for (N, level, H, W, K, seed) in [(3, 3, 64, 64, 50, 1), (2, 2, 24, 40, 7, 2)]:
        fv_np, fv, first, nf = _inputs(N, level, seed)
        assert native.workspace_bytes(fv.shape[0], N, H, W, K) <= native.STATIC_SCRATCH_BYTES
        want = native.rasterize_meshes(fv, first, nf, (H, W), blur, K, True)
        out = (torch.full((N, H, W, K), -7, dtype=torch.int32, device="cuda"),
               torch.full((N, H, W, K), -7.0, device="cuda"), torch.full((N, H, W, K, 3), -7.0, device="cuda"),
               torch.full((N, H, W, K), -7.0, device="cuda"))
        torch.cuda.synchronize()
        before = torch.cuda.memory_allocated()
        native.rasterize_meshes_static(fv, first, nf, (H, W), blur, K, True, out)
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated() == before
        for a, b in zip(out, want):
            assert torch.equal(a, b)
        # and against the oracle directly (the bar of the workspace entry point)
        ref = orc.rasterize(fv_np, N, H, W, blur, K, clip_bary=True)
        np.testing.assert_array_equal(out[0].cpu().numpy(), ref[0].astype(np.int32))
        assert (ref[0] >= 0).sum() > 0


def test_static_scratch_rejects_what_does_not_fit():
    from unicorn_amd import native
    N, H, W, K = 64, 128, 128, 50  # K-list spill alone is far beyond 64 MiB
    fv_np, fv, first, nf = _inputs(1, 1, 3)
    F = fv.shape[0]
    assert native.workspace_bytes(N * F, N, H, W, K) > native.STATIC_SCRATCH_BYTES
    fvN = fv.repeat(N, 1, 1)
    firstN = torch.arange(N, dtype=torch.int64, device="cuda") * F
    nfN = torch.full((N,), F, dtype=torch.int64, device="cuda")
    out = (torch.empty((N, H, W, K), dtype=torch.int32, device="cuda"), None, None, None)
    with pytest.raises(RuntimeError, match="static scratch"):
        native.rasterize_meshes_static(fvN, firstN, nfN, (H, W), orc.blur_radius_for(1e-4), K, True, out)
