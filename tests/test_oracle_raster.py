"""Known-answer and finite-difference tests that anchor the rasterizer part of the oracle.

PARITY UNPINNED: pytorch3d 0.6.2 (environment.yml:21) is absent from /root/reference and from this image and the
reference holds no golden vectors for it, so these tests check the restated algorithm (SURVEY.md 8a spec block)
against closed-form answers and float64 finite differences instead."""
import numpy as np
import pytest

from oracle import orc
from unicorn_amd import synth


def ndc(i, S):
    """pixel-centre NDC coordinate of output row/col i (+X left, +Y up)"""
    return -1.0 + (2 * (S - 1 - i) + 1) / S


def tri(v0, v1, v2):
    return np.array([[v0, v1, v2]], np.float64)


def seg_d2(p, a, b):
    a, b, p = map(np.asarray, (a, b, p))
    t = np.clip(np.dot(b - a, p - a) / np.dot(b - a, b - a), 0, 1)
    return float(np.sum((p - (a + t * (b - a))) ** 2))


def test_single_triangle_closed_form():
    H = W = 8
    fv = tri([-0.7, -0.6, 1.0], [0.8, -0.5, 2.0], [0.1, 0.7, 3.0])
    p2f, zb, bar, ds = orc.rasterize(fv, 1, H, W, 0.0, 1, dtype=np.float64)
    A = np.array([[fv[0, 0, 0], fv[0, 1, 0], fv[0, 2, 0]], [fv[0, 0, 1], fv[0, 1, 1], fv[0, 2, 1]], [1, 1, 1]])
    n_in = 0
    for yo in range(H):
        for xo in range(W):
            p = np.array([ndc(xo, W), ndc(yo, H)])
            w = np.linalg.solve(A, np.array([p[0], p[1], 1.0]))
            if (w > 0).all():
                n_in += 1
                assert p2f[0, yo, xo, 0] == 0
                np.testing.assert_allclose(bar[0, yo, xo, 0], w, atol=1e-6)
                np.testing.assert_allclose(zb[0, yo, xo, 0], w @ fv[0, :, 2], atol=1e-6)
                d = min(seg_d2(p, fv[0, i, :2], fv[0, j, :2]) for i, j in [(0, 1), (0, 2), (1, 2)])
                np.testing.assert_allclose(ds[0, yo, xo, 0], -d, atol=1e-9)
            else:
                assert p2f[0, yo, xo, 0] == -1
                assert zb[0, yo, xo, 0] == -1 and ds[0, yo, xo, 0] == -1 and (bar[0, yo, xo, 0] == -1).all()
    assert n_in > 5


def test_pixel_on_edge_is_not_inside():
    # pixel centres of an 4x4 image sit at +-0.25, +-0.75; the edge x = 0.25 passes exactly through a column
    H = W = 4
    fv = tri([0.25, -2.0, 1.0], [0.25, 2.0, 1.0], [-3.0, 0.0, 1.0])
    p2f, _, _, ds = orc.rasterize(fv, 1, H, W, 0.0, 1, dtype=np.float32)
    col_on_edge = [xo for xo in range(W) if ndc(xo, W) == 0.25][0]
    assert (p2f[0, :, col_on_edge, 0] == -1).all()  # strict '>' on the barycentrics
    assert (p2f[0, :, col_on_edge + 1:, 0] == 0).all()  # +X is left: larger xo = smaller x = inside
    # with a blur halo the on-edge pixel is kept, with distance 0 and NOT inside
    p2f, _, _, ds = orc.rasterize(fv, 1, H, W, 1e-3, 1, dtype=np.float32)
    assert (p2f[0, :, col_on_edge, 0] == 0).all() and (ds[0, :, col_on_edge, 0] == 0).all()


def test_depth_order_tie_break_and_topk():
    H = W = 4
    big = [[-3.0, -3.0], [3.0, -3.0], [0.0, 3.0]]
    zs = [5.0, 2.0, 2.0, 9.0, 1.0]
    fv = np.array([[[x, y, z] for x, y in big] for z in zs])
    p2f, zb, _, _ = orc.rasterize(fv, 1, H, W, 0.0, 3, dtype=np.float32)
    assert (p2f[0, 1, 1] == [4, 1, 2]).all()  # z=1, then the z=2 tie in face-index order; z=5 and 9 dropped
    np.testing.assert_array_equal(zb[0, 1, 1], [1, 2, 2])
    p2f, zb, _, _ = orc.rasterize(fv, 1, H, W, 0.0, 8, dtype=np.float32)
    assert (p2f[0, 2, 2] == [4, 1, 2, 0, 3, -1, -1, -1]).all()


def test_behind_camera_and_degenerate_faces_are_culled():
    H = W = 4
    big = [[-3.0, -3.0], [3.0, -3.0], [0.0, 3.0]]
    behind = [[x, y, -1.0] for x, y in big]
    degenerate = [[-1.0, -1.0, 1.0], [0.0, 0.0, 1.0], [1.0, 1.0, 1.0]]  # zero area
    ok = [[x, y, 4.0] for x, y in big]
    p2f, _, _, _ = orc.rasterize(np.array([behind, degenerate, ok]), 1, H, W, 1e-3, 3, dtype=np.float32)
    assert (p2f[..., 0] == 2).all() and (p2f[..., 1:] == -1).all()
    # a face whose interpolated depth at the pixel is negative is skipped even if one vertex is in front
    straddle = [[-3.0, -3.0, -5.0], [3.0, -3.0, -5.0], [0.0, 3.0, 0.5]]
    p2f, zb, _, _ = orc.rasterize(np.array([straddle]), 1, H, W, 0.0, 1, dtype=np.float32)
    assert ((p2f == -1) | (zb >= 0)).all() and (p2f == -1).any()


def test_blur_halo_threshold():
    # edge x = 0 ; pixel centres at x = +-1/8, +-3/8 ... for W = 8 -> squared distances 1/64, 9/64
    H = W = 8
    fv = tri([0.0, -5.0, 1.0], [0.0, 5.0, 1.0], [-9.0, 0.0, 1.0])  # covers x < 0
    for blur, expect_halo in [(1.0 / 64 + 1e-6, True), (1.0 / 64, False), (1.0 / 64 - 1e-6, False)]:
        p2f, _, _, ds = orc.rasterize(fv, 1, H, W, blur, 1, dtype=np.float64)
        xo = [x for x in range(W) if abs(ndc(x, W) - 0.125) < 1e-12][0]
        assert (p2f[0, :, xo, 0] == (0 if expect_halo else -1)).all()  # kept iff dist < blur_radius (strict)
        if expect_halo:
            np.testing.assert_allclose(ds[0, :, xo, 0], 1.0 / 64, atol=1e-12)  # outside -> positive


def test_clipped_barycentrics_outside_face():
    H = W = 8
    fv = tri([0.0, -5.0, 1.0], [0.0, 5.0, 2.0], [-9.0, 0.0, 3.0])
    p2f, zb, bar, _ = orc.rasterize(fv, 1, H, W, 0.05, 1, clip_bary=True, dtype=np.float64)
    sel = p2f[..., 0] == 0
    b = bar[..., 0, :][sel]
    assert (b >= 0).all()
    np.testing.assert_allclose(b.sum(-1), 1.0, atol=1e-9)
    np.testing.assert_allclose(zb[..., 0][sel], b @ np.array([1.0, 2.0, 3.0]), atol=1e-9)
    # unclipped mode: halo pixels carry a negative barycentric
    _, _, bar_u, _ = orc.rasterize(fv, 1, H, W, 0.05, 1, clip_bary=False, dtype=np.float64)
    assert (bar_u[..., 0, :][sel] < 0).any()


def test_packed_face_indices_and_batch_independence():
    sc = synth.make_scene(3, level=1, seed=3)
    s = 3.7320508
    vn = orc.project_verts(sc["verts"].numpy(), sc["R"].numpy(), sc["T"].numpy(), s)
    F = sc["faces"].shape[0]
    fv = np.stack([vn[n][sc["faces"].numpy()] for n in range(3)]).reshape(-1, 3, 3)
    p2f, zb, bar, ds = orc.rasterize(fv, 3, 16, 16, orc.blur_radius_for(1e-3), 6)
    for n in range(3):
        v = p2f[n][p2f[n] >= 0]
        assert v.size and v.min() >= n * F and v.max() < (n + 1) * F
        one = orc.rasterize(fv[n * F:(n + 1) * F], 1, 16, 16, orc.blur_radius_for(1e-3), 6)
        np.testing.assert_array_equal(np.where(p2f[n] >= 0, p2f[n] - n * F, -1), one[0][0])
        np.testing.assert_array_equal(zb[n], one[1][0])
    # sorted ascending in z, padding last
    z = np.where(p2f >= 0, zb, 1e30)
    assert (np.diff(z, axis=-1) >= 0).all()


def test_tile_lists_cover_naive_hits():
    sc = synth.make_scene(2, level=2, seed=5)
    vn = orc.project_verts(sc["verts"].numpy(), sc["R"].numpy(), sc["T"].numpy(), 3.7320508)
    faces = sc["faces"].numpy()
    F = faces.shape[0]
    fv = np.stack([vn[n][faces] for n in range(2)]).reshape(-1, 3, 3)
    H, W, blur = 24, 20, orc.blur_radius_for(3e-4)  # ragged: not multiples of 8
    p2f = orc.rasterize(fv, 2, H, W, blur, 40)[0]
    assert (p2f[..., -1] == -1).all()  # K large enough to hold every hit
    counts, lists = orc.tile_faces(fv, 2, H, W, blur, 8, 2 * F)
    assert counts.max() <= 2 * F and counts.max() > 0
    for n in range(2):
        for ty in range(counts.shape[1]):
            for tx in range(counts.shape[2]):
                lst = lists[n, ty, tx, :counts[n, ty, tx]]
                assert (np.diff(lst) > 0).all()
                hits = p2f[n, ty * 8:(ty + 1) * 8, tx * 8:(tx + 1) * 8]
                assert set(hits[hits >= 0].tolist()) <= set(lst.tolist())


def _fd(fun, x, eps):
    g = np.zeros_like(x)
    flat, gf = x.reshape(-1), g.reshape(-1)
    for i in range(flat.size):
        old = flat[i]
        flat[i] = old + eps
        fp = fun()
        flat[i] = old - eps
        fm = fun()
        flat[i] = old
        gf[i] = (fp - fm) / (2 * eps)
    return g


@pytest.mark.parametrize("clip", [True, False])
def test_rasterize_backward_finite_differences(clip):
    rs = np.random.RandomState(0)
    fv = np.array([[[-0.6, -0.5, 1.0], [0.7, -0.4, 1.5], [0.0, 0.8, 2.0]],
                   [[-0.3, -0.7, 1.2], [0.9, 0.1, 1.1], [-0.5, 0.6, 1.3]]]) + rs.randn(2, 3, 3) * 0.01
    H, W, K, blur = 6, 6, 2, 0.02
    p2f, zb, bar, ds = orc.rasterize(fv, 1, H, W, blur, K, clip_bary=clip, dtype=np.float64)
    valid = p2f >= 0
    assert valid.sum() > 20
    cz, cb, cd = rs.randn(*zb.shape), rs.randn(*bar.shape), rs.randn(*ds.shape)

    def loss():
        p, z, b, d = orc.rasterize(fv, 1, H, W, blur, K, clip_bary=clip, dtype=np.float64)
        assert (p == p2f).all()  # the perturbation must not change visibility
        return float((z * cz)[valid].sum() + (b * cb)[valid].sum() + (d * cd)[valid].sum())

    g = orc.rasterize_backward(fv, p2f, cz * valid, cb * valid[..., None], cd * valid, clip, dtype=np.float64)
    g_fd = _fd(loss, fv, 1e-7)
    np.testing.assert_allclose(g, g_fd.reshape(g.shape), rtol=1e-5, atol=1e-6)


def test_render_end_to_end_finite_differences():
    sc = synth.make_scene(1, level=0, tex_size=4, jitter=0.0, seed=2)  # icosahedron: 12 verts / 20 faces
    verts, faces = sc["verts"].double().numpy(), sc["faces"].numpy()
    R, T, maps = sc["R"].double().numpy(), sc["T"].double().numpy(), sc["maps"].double().numpy()
    uvs = sc["verts_uvs"][None].double().numpy()
    H = W = 6
    sigma = 3e-3  # wide soft edges so that FD steps stay on one smooth piece
    cot = np.random.RandomState(1).randn(1, 4, H, W)
    kw = dict(H=H, W=W, K=8, s=3.7320508, sigma=sigma, dtype=np.float64)

    def loss():
        return float((orc.render_fwd_bwd(verts, faces, R, T, maps, uvs, **kw) * cot).sum())

    img, (gv, gR, gT, gm) = orc.render_fwd_bwd(verts, faces, R, T, maps, uvs, grad_image=cot, **kw)
    assert 0.05 < img[0, 3].mean() < 0.95
    np.testing.assert_allclose(gm, _fd(loss, maps, 1e-6), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(gT, _fd(loss, T, 1e-7), rtol=2e-4, atol=1e-4)
    np.testing.assert_allclose(gv, _fd(loss, verts, 1e-7), rtol=2e-4, atol=1e-4)
    np.testing.assert_allclose(gR, _fd(loss, R, 1e-7), rtol=2e-4, atol=1e-4)


def test_render_multithreaded_equals_single():
    sc = synth.make_scene(4, level=1, tex_size=8, seed=4)
    args = (sc["verts"].numpy(), sc["faces"].numpy(), sc["R"].numpy(), sc["T"].numpy(), sc["maps"].numpy(),
            sc["verts_uvs"][None].expand(4, -1, -1).numpy())
    cot = np.random.RandomState(2).randn(4, 4, 16, 16).astype(np.float32)
    a = orc.render_fwd_bwd(*args, H=16, W=16, K=10, s=3.73, grad_image=cot, n_threads=1)
    b = orc.render_fwd_bwd(*args, H=16, W=16, K=10, s=3.73, grad_image=cot, n_threads=3)
    np.testing.assert_array_equal(a[0], b[0])
    for x, y in zip(a[1], b[1]):
        np.testing.assert_array_equal(x, y)
