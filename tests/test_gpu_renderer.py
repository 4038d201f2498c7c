"""End-to-end GPU parity of the `Renderer` drop-in (reference API: src/model/renderer.py:25-116) against the CPU
oracle's whole-path restatement, at the BASELINE.json configurations (batch reduced where the oracle would be slow)."""
import numpy as np
import pytest
import torch

from oracle import orc
from unicorn_amd import synth

pytestmark = pytest.mark.gpu


def _setup(N, level, img, K, seed=0, sigma=1e-4, tex=64, cam=None, expand=True, fused=True, **rkw):
    from unicorn_amd.renderer import Renderer
    from unicorn_amd.structures import Meshes, TexturesUV
    dev = torch.device("cuda:0")
    sc = synth.make_scene(N, level=level, tex_size=tex, seed=seed)
    cam = cam or {"name": "fov", "fov": 30}
    rend = Renderer(img, cameras=cam, faces_per_pixel=K, sigma=sigma, **rkw).to(dev)
    rend.renderer.fused = rend.viz_renderer.fused = fused
    verts = sc["verts"].to(dev).requires_grad_(True)
    maps = sc["maps"].to(dev).requires_grad_(True)
    R = sc["R"].to(dev).requires_grad_(True)
    T = sc["T"].to(dev).requires_grad_(True)
    faces, uvs = sc["faces"].to(dev), sc["verts_uvs"].to(dev)
    if expand:  # unicorn.py:283
        fb, ub = faces[None].expand(N, -1, -1), uvs[None].expand(N, -1, -1)
    else:
        fb, ub = faces[None].repeat(N, 1, 1), uvs[None].repeat(N, 1, 1)
    meshes = Meshes(verts=verts, faces=fb, textures=TexturesUV(maps, fb, ub))
    return sc, rend, meshes, (verts, maps, R, T)


def _oracle(sc, H, W, K, s, sigma, cot=None, **kw):
    N = sc["verts"].shape[0]
    return orc.render_fwd_bwd(sc["verts"].numpy(), sc["faces"].numpy(), sc["R"].numpy(), sc["T"].numpy(),
                              sc["maps"].numpy(), sc["verts_uvs"][None].expand(N, -1, -1).numpy(), H, W, K, s,
                              sigma=sigma, grad_image=cot, **kw)


def rel_err(a, b):
    return float(np.abs(a - b).max() / max(float(np.abs(b).max()), 1e-20))


S30 = float(1.0 / np.tan(np.deg2rad(15.0)))


def assert_grads_to_the_bar(got, g32, g64_fn, what=""):
    """north_star's bar: gradients within 1e-4 (relative) of the float32 oracle.  A gradient over that bar must be
    SHOWN ill-conditioned, not waved through: the float32 oracle itself is then that far from the float64 oracle
    (`noise`), and the GPU is held to the float64 answer within 1e-4 + 30 x noise -- the criterion of
    tests/campaign_random_parity.py (clip-renormalisation residues on sliver faces).
    got / g32: {name: array}; g64_fn() -> {name: array}, the float64 oracle, only evaluated when needed."""
    g64 = None
    for name, a in got.items():
        e = rel_err(a, g32[name])
        if e <= 1e-4:
            continue
        if g64 is None:
            g64 = g64_fn()
        noise, e64 = rel_err(g32[name], g64[name]), rel_err(a, g64[name])
        assert e64 <= 1e-4 + 30.0 * noise, (f"{what} grad {name}: {e:g} from the float32 oracle, {e64:g} from the float64 "
                                            f"oracle, whose distance from the float32 oracle is {noise:g}")

CASES = [
    # N, level, img, K, camera
    (4, 3, 64, 50, None),                                                          # configs[1]: sn, K=50
    (4, 3, 64, 25, None),                                                          # configs[2]: car.yml, K=25
    (2, 3, 128, 50, None),                                                         # headline: 128^2, K=50
    (1, 4, 128, 50, {"name": "perspective", "focal_length": 3.732050895690918}),   # configs[3]: finer mesh, sn camera
    (2, 2, (40, 56), 10, None),                                                    # non-square, ragged tiles
    (2, 3, 64, 3, None),                                                           # tiny K: lists fill up, the exact
    (2, 3, (33, 70), 1, None),                                                     # keep-the-K-smallest route is taken
    (1, 5, 64, 20, None),                                                          # 20480 faces: ten scan rounds
]


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "two-stage"])
@pytest.mark.parametrize("N,level,img,K,cam", CASES)
def test_forward_backward_match_oracle(N, level, img, K, cam, fused):
    sc, rend, meshes, (verts, maps, R, T) = _setup(N, level, img, K, seed=K + level, cam=cam, fused=fused)
    H, W = (img, img) if isinstance(img, int) else img
    out = rend(meshes, R, T)
    assert out.shape == (N, 4, H, W) and out.dtype == torch.float32
    cot = np.random.RandomState(0).randn(N, 4, H, W).astype(np.float32)
    (out * torch.from_numpy(cot).cuda()).sum().backward()
    s = S30 if cam is None else cam["focal_length"]
    ref, (gv, gR, gT, gm) = _oracle(sc, H, W, K, s, 1e-4, cot)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, atol=1e-5, rtol=0)
    assert 0.05 < ref[:, 3].mean() < 0.9
    assert rel_err(maps.grad.cpu().numpy(), gm) <= 1e-4
    assert rel_err(verts.grad.cpu().numpy(), gv) <= 1e-4
    assert rel_err(T.grad.cpu().numpy(), gT) <= 1e-4
    assert rel_err(R.grad.cpu().numpy(), gR) <= 1e-4


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "two-stage"])
def test_sigmoid_blend_and_background(fused):
    sc, rend, meshes, (verts, maps, R, T) = _setup(2, 2, 32, 12, seed=3, sigma=1e-3, clip_inside=False, fused=fused,
                                                   background_color=(0.1, 0.6, 0.3))
    out = rend(meshes, R, T)
    cot = np.random.RandomState(1).randn(*out.shape).astype(np.float32)
    (out * torch.from_numpy(cot).cuda()).sum().backward()
    ref, (gv, gR, gT, gm) = _oracle(sc, 32, 32, 12, S30, 1e-3, cot, clip_inside=False, background=(0.1, 0.6, 0.3))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, atol=1e-5, rtol=0)
    assert rel_err(verts.grad.cpu().numpy(), gv) <= 1e-4 and rel_err(maps.grad.cpu().numpy(), gm) <= 1e-4


def test_only_requested_gradients_are_produced():
    # pose step: meshes detached, grads to R, T only (unicorn.py:227-228); 3D step: R, T detached (:230)
    sc, rend, meshes, (verts, maps, R, T) = _setup(2, 2, 32, 10, seed=5)
    out = rend(meshes.detach(), R, T)
    out.sum().backward()
    assert verts.grad is None and maps.grad is None and R.grad is not None and T.grad is not None
    gR = R.grad.clone()
    R.grad = T.grad = None
    out = rend(meshes, R.detach(), T.detach())
    out.sum().backward()
    assert verts.grad is not None and maps.grad is not None and R.grad is None
    # same numbers as when everything is differentiated at once
    sc2, rend2, meshes2, (v2, m2, R2, T2) = _setup(2, 2, 32, 10, seed=5)
    rend2(meshes2, R2, T2).sum().backward()
    assert rel_err(gR.cpu().numpy(), R2.grad.cpu().numpy()) <= 1e-5
    assert rel_err(verts.grad.cpu().numpy(), v2.grad.cpu().numpy()) <= 1e-5


def test_repeat_tiles_candidates_k_major():
    """utils/mesh.py:43-71 + unicorn.py:232-233: K pose candidates per mesh, K-major."""
    from unicorn_amd.structures import repeat
    B, KP = 2, 3
    sc, rend, meshes, (verts, maps, R, T) = _setup(B, 2, 32, 10, seed=7)
    sc6 = synth.make_scene(B * KP, level=2, seed=11)
    R6, T6 = sc6["R"].cuda(), sc6["T"].cuda()
    out = rend(repeat(meshes, KP), R6, T6)
    for k in range(KP):
        one = rend(meshes, R6[k * B:(k + 1) * B], T6[k * B:(k + 1) * B])
        assert torch.equal(out[k * B:(k + 1) * B], one)
    out.sum().backward()
    assert verts.grad.shape == (B, verts.shape[1], 3) and maps.grad.abs().sum() > 0


@pytest.mark.parametrize("shared_tables", [True, False], ids=["shared-topology", "per-mesh-tables"])
def test_tiled_poses_equal_materialised_copies(shared_tables):
    """utils.mesh.repeat (utils/mesh.py:43-71) as an index instead of copies -- ur_render_meshes_* with
    n_meshes = B reads mesh n % B for image n and sums a mesh's gradients over its poses.  Against the same batch
    physically tiled: images bit-identical, gradients equal to rounding; and nothing was copied on the way."""
    from unicorn_amd.structures import Meshes, TexturesUV, repeat
    B, KP = 3, 4
    sc, rend, meshes, (verts, maps, R, T) = _setup(B, 2, 32, 10, seed=17, expand=shared_tables)
    if not shared_tables:  # every mesh its own UV table as well
        tex = meshes.textures
        ub = tex.verts_uvs_padded() * torch.tensor([1.0, 0.9, 0.8], device="cuda").view(B, 1, 1)
        meshes.textures = TexturesUV(maps, tex.faces_uvs_padded(), ub)
    poses = synth.make_scene(B * KP, level=0, tex_size=2, seed=3)
    R6, T6 = poses["R"].cuda().requires_grad_(True), poses["T"].cuda().requires_grad_(True)
    tiled = repeat(meshes, KP)
    out = rend(tiled, R6, T6)
    assert tiled._tiled_verts is None and tiled.textures._tiled is None        # no copy of verts / maps was made
    cot = torch.randn(out.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    out.backward(cot)
    got = [t.grad.clone() for t in (verts, maps, R6, T6)]
    assert got[0].shape == verts.shape and got[1].shape == maps.shape
    for t in (verts, maps, R6, T6):
        t.grad = None
    tex = meshes.textures
    rep3 = lambda t: t.repeat(KP, 1, 1)
    copies = Meshes(rep3(verts), rep3(meshes.faces_padded()),
                    TexturesUV(maps.repeat(KP, 1, 1, 1), rep3(tex.faces_uvs_padded()), rep3(tex.verts_uvs_padded())))
    out2 = rend(copies, R6, T6)
    assert torch.equal(out, out2)
    out2.backward(cot)
    for a, t in zip(got, (verts, maps, R6, T6)):
        assert rel_err(a.cpu().numpy(), t.grad.cpu().numpy()) <= 1e-5
    # the two-stage path (fragments) takes the tiled batch as an ordinary one
    rend.renderer.fused = False
    assert torch.equal(rend(repeat(meshes, KP), R6, T6), out)


def test_tiling_is_refused_where_the_kernels_cannot_index():
    """n_meshes that does not divide the poses -> error, not a wrong picture"""
    from unicorn_amd import native
    assert native.can_tile(642, 1280, 64, 64) and not native.can_tile(642, 40000, 64, 64)
    sc, rend, meshes, (verts, maps, R, T) = _setup(2, 1, 16, 4, seed=1)
    from unicorn_amd import ops
    fu, shared = meshes.textures.faces_verts_uvs()
    first, nf = rend.renderer.rasterizer._mesh_index(3, sc["faces"].shape[0], verts.device)
    R3, T3 = R.detach()[[0, 1, 0]].contiguous(), T.detach()[[0, 1, 0]].contiguous()
    with pytest.raises(RuntimeError, match="does not divide"):
        ops.render_meshes(verts.detach(), maps.detach(), R3, T3, sc["faces"].cuda(), S30, S30, first, nf, (16, 16),
                          1e-3, 4, True, fu, shared, 1e-4, (1.0, 1.0, 1.0), None, 1)


def test_list_constructed_meshes_equal_padded():
    """the reference's repeat() builds Meshes from python lists of cloned tensors"""
    from unicorn_amd.structures import Meshes, TexturesUV
    sc, rend, meshes, (verts, maps, R, T) = _setup(3, 2, 32, 10, seed=9)
    faces = sc["faces"].cuda()
    m2 = Meshes([v.clone() for v in verts.detach()], [faces.clone() for _ in range(3)],
                TexturesUV(maps.detach(), faces[None].repeat(3, 1, 1), sc["verts_uvs"].cuda()[None].repeat(3, 1, 1)))
    assert torch.equal(rend(meshes, R, T), rend(m2, R, T))


def test_viz_path_matches_oracle():
    """renderer.py:51-57,119-124: hard raster at 2x viz_size, K=1, sigma=0 blend, 2x2 average pool; no grad."""
    sc, rend, meshes, (verts, maps, R, T) = _setup(2, 3, 64, 25, seed=4, viz_size=(48, 48))
    out = rend(meshes, R, T, viz_purpose=True)
    assert out.shape == (2, 4, 48, 48) and not out.requires_grad
    ref = _oracle(sc, 96, 96, 1, S30, 0.0, blur_radius=0.0)
    ref = ref.reshape(2, 4, 48, 2, 48, 2).mean((3, 5))
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=1e-5, rtol=0)


def test_compute_vertex_visibility():
    sc, rend, meshes, (verts, maps, R, T) = _setup(2, 2, 32, 5, seed=6, viz_size=(32, 32))
    vis = rend.compute_vertex_visibility(meshes, R, T)
    N, V = verts.shape[:2]
    assert vis.shape == (N, V) and vis.dtype == torch.bool
    vn = orc.project_verts(sc["verts"].numpy(), sc["R"].numpy(), sc["T"].numpy(), S30)
    faces = sc["faces"].numpy()
    fv = np.stack([vn[n][faces] for n in range(N)]).reshape(-1, 3, 3)
    p2f = orc.rasterize(fv, N, 64, 64, 0.0, 1)[0]
    expect = np.zeros((N, V), bool)
    F = faces.shape[0]
    for f in np.unique(p2f[p2f >= 0]):
        expect[f // F, faces[f % F]] = True
    np.testing.assert_array_equal(vis.cpu().numpy(), expect)
    assert 0.2 < expect.mean() < 0.8  # roughly the camera-facing half


def test_fragments_surface():
    sc, rend, meshes, (verts, maps, R, T) = _setup(2, 2, 32, 5, seed=6)
    frag = rend.renderer.rasterizer(meshes, R=R, T=T)
    assert frag.pix_to_face.dtype == torch.int64 and frag.pix_to_face.shape == (2, 32, 32, 5)
    assert frag.bary_coords.shape == (2, 32, 32, 5, 3) and frag.zbuf.shape == frag.dists.shape == (2, 32, 32, 5)
    tex = meshes.sample_textures(frag)  # (N,H,W,K,3) texels, as pytorch3d returns them
    vn = orc.project_verts(sc["verts"].numpy(), sc["R"].numpy(), sc["T"].numpy(), S30)
    faces = sc["faces"].numpy()
    fv = np.stack([vn[n][faces] for n in range(2)]).reshape(-1, 3, 3)
    p2f, zb, bar, ds = orc.rasterize(fv, 2, 32, 32, orc.blur_radius_for(1e-4), 5)
    fuv = np.tile(sc["verts_uvs"].numpy()[faces][None], (2, 1, 1, 1)).reshape(-1, 3, 2)
    np.testing.assert_allclose(tex.detach().cpu().numpy(), orc.sample_textures(sc["maps"].numpy(), fuv, p2f, bar),
                               atol=1e-5)


def test_init_kwargs_roundtrip_and_unsupported():
    from unicorn_amd.renderer import Renderer
    kw = {"cameras": {"name": "perspective", "focal_length": 3.73}, "faces_per_pixel": 50, "sigma": 1e-4}
    r = Renderer(64, **kw)
    assert r.init_kwargs == kw and r.init_kwargs is not r.init_kwargs
    assert Renderer(64, **r.init_kwargs).renderer.rasterizer.raster_settings.faces_per_pixel == 50
    with pytest.raises(NotImplementedError):
        Renderer(64, shading_type="toon")


def test_fused_equals_two_stage_bitwise_forward():
    a = _setup(3, 3, 64, 50, seed=21, fused=True)
    b = _setup(3, 3, 64, 50, seed=21, fused=False)
    assert torch.equal(a[1](a[2], a[3][2], a[3][3]), b[1](b[2], b[3][2], b[3][3]))


@pytest.mark.parametrize("K,level,H", [(30, 2, 24), (100, 3, 24), (150, 3, 16)], ids=["K30", "K100", "K150"])
@pytest.mark.parametrize("mode_kw", [dict(clip_inside=False), dict(clip_inside=True)], ids=["sigmoid", "clip"])
def test_deep_translucent_stacks(mode_kw, K, level, H):
    """wide soft edges: many translucent layers per pixel (more than the backward keeps in LDS at once).  K30: lists
    of up to 60 candidates cut to 30; K100: the K > 64 build, lists of 65..128 entries (two entries per lane in the
    sort) that fill up; K150: lists longer than any sorting network (in-place compaction + per-lane selection sort)"""
    sigma = 3e-2
    sc, rend, meshes, (verts, maps, R, T) = _setup(2, level, H, K, seed=13, sigma=sigma, **mode_kw)
    out = rend(meshes, R, T)
    cot = np.random.RandomState(5).randn(*out.shape).astype(np.float32)
    (out * torch.from_numpy(cot).cuda()).sum().backward()
    ref, (gv, gR, gT, gm) = _oracle(sc, H, H, K, S30, sigma, cot, **mode_kw)
    frag = rend.renderer.rasterizer(meshes, R=R, T=T)
    assert ((frag.pix_to_face >= 0).sum(-1) > 16).float().mean() > 0.3  # the stacks really are deep
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, atol=1e-5, rtol=0)
    assert rel_err(maps.grad.cpu().numpy(), gm) <= 1e-4
    assert rel_err(verts.grad.cpu().numpy(), gv) <= 1e-4
    assert rel_err(T.grad.cpu().numpy(), gT) <= 1e-4


def test_pixel_centres_exactly_on_edges():
    """axis-aligned quad whose edges run through pixel centres: dist == 0 with alpha == 1 but d(alpha)/d(dist) != 0,
    the one case where the layers BEHIND an opaque layer matter to the gradient"""
    from unicorn_amd.renderer import Renderer
    from unicorn_amd.structures import Meshes, TexturesUV
    dev = torch.device("cuda:0")
    H = 8
    s = 1.0
    # two stacked quads (4 faces) at depth 2 and 3, in NDC-friendly world coords: camera = identity, focal 1
    def quad(z, e):
        return [[-e * z, -e * z, z], [e * z, -e * z, z], [e * z, e * z, z], [-e * z, e * z, z]]
    verts = torch.tensor([quad(2.0, 0.375) + quad(3.0, 0.625)], dtype=torch.float32)  # edges at +-0.375, +-0.625
    faces = torch.tensor([[0, 1, 2], [0, 2, 3], [4, 5, 6], [4, 6, 7]])
    uvs = torch.rand(8, 2, generator=torch.Generator().manual_seed(0))
    maps = torch.rand(1, 8, 8, 3, generator=torch.Generator().manual_seed(1))
    R, T = torch.eye(3)[None], torch.zeros(1, 3)
    rend = Renderer(H, cameras={"name": "perspective", "focal_length": s}, faces_per_pixel=6, sigma=1e-4).to(dev)
    v, m = verts.to(dev).requires_grad_(True), maps.to(dev).requires_grad_(True)
    meshes = Meshes(v, faces[None].to(dev), TexturesUV(m, faces[None].to(dev), uvs[None].to(dev)))
    out = rend(meshes, R.to(dev), T.to(dev))
    cot = np.random.RandomState(2).randn(1, 4, H, H).astype(np.float32)
    (out * torch.from_numpy(cot).cuda()).sum().backward()
    ref, (gv, gR, gT, gm) = orc.render_fwd_bwd(verts.numpy(), faces.numpy(), R.numpy(), T.numpy(), maps.numpy(),
                                               uvs[None].numpy(), H, H, 6, s, grad_image=cot)
    fv = orc.project_verts(verts.numpy(), R.numpy(), T.numpy(), s)[0][faces.numpy()]
    ds = orc.rasterize(fv, 1, H, H, orc.blur_radius_for(1e-4), 6)[3]
    assert (ds == 0).sum() > 0  # the case is really exercised
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, atol=1e-5, rtol=0)
    assert rel_err(v.grad.cpu().numpy(), gv) <= 1e-4
    assert rel_err(m.grad.cpu().numpy(), gm) <= 1e-4


def test_graphed_step_matches_eager():
    """hipGraph replay of a whole forward + backward (unicorn_amd/graphs.py) gives the eager image and gradients"""
    from unicorn_amd.graphs import GraphedStep
    sc, rend, meshes, leaves = _setup(4, 3, 64, 25)
    verts, maps, R, T = leaves
    cot = torch.randn(4, 4, 64, 64, device=verts.device, generator=torch.Generator(device=verts.device).manual_seed(1))

    def body():
        img = rend(meshes, R, T)
        img.backward(cot)
        return img

    img_e = body().detach().clone()
    grads_e = [t.grad.detach().clone() for t in leaves]
    step = GraphedStep(body, grads_of=leaves)
    for _ in range(2):  # replays rewrite the same static tensors
        img_g = step()
    torch.cuda.synchronize()
    assert torch.equal(img_g.detach(), img_e)  # the forward is deterministic
    for t, ge in zip(leaves, grads_e):
        assert rel_err(t.grad.cpu().numpy(), ge.cpu().numpy()) < 1e-5  # atomics: last-place differences only


def test_kernel_timing_hook():
    """ur_profile_begin / ur_profile_end bracket the two dominant kernels with HIP events (bench.py's roofline leg)"""
    from unicorn_amd import native
    sc, rend, meshes, leaves = _setup(2, 3, 64, 25)
    verts, maps, R, T = leaves
    native.profile_begin(8)
    for _ in range(3):
        rend(meshes, R, T).sum().backward()
    torch.cuda.synchronize()
    got = native.profile_end(8)
    assert len(got["k_raster<2>"]) == 3 and len(got["k_render_bwd"]) == 3
    assert all(0.0 < ms < 50.0 for v in got.values() for ms in v)
    with pytest.raises(RuntimeError):
        native.profile_end(8)  # not on any more


def _soup(N, F, seed, behind=True):
    """random triangle soup: sizes from sub-pixel to screen-filling, a few degenerate faces, a few crossing or behind
    the camera plane; view-space coordinates (identity camera rotation, zero translation)"""
    rs = np.random.RandomState(seed)
    V = 3 * F
    centre = np.concatenate([rs.uniform(-0.8, 0.8, (N, F, 1, 2)), rs.uniform(1.5, 4.0, (N, F, 1, 1))], -1)
    size = np.exp(rs.uniform(np.log(0.005), np.log(1.5), (N, F, 1, 1)))
    verts = (centre + size * rs.randn(N, F, 3, 3) * np.array([1, 1, 0.3])).astype(np.float32)
    verts[:, 0, 1] = verts[:, 0, 0]                      # degenerate: two corners coincide
    verts[:, 1, 2] = 0.5 * (verts[:, 1, 0] + verts[:, 1, 1])  # degenerate: collinear
    if behind:
        verts[:, 2, :, 2] = -1.0                          # fully behind the camera
        verts[:, 3, 0, 2] = -0.5                          # crossing the camera plane
    faces = np.arange(V).reshape(F, 3)
    uvs = rs.rand(V, 2).astype(np.float32)
    maps = rs.rand(N, 16, 16, 3).astype(np.float32)
    R = np.tile(np.eye(3, dtype=np.float32), (N, 1, 1))
    T = np.zeros((N, 3), np.float32)
    return dict(verts=torch.tensor(verts.reshape(N, V, 3)), faces=torch.tensor(faces), verts_uvs=torch.tensor(uvs),
                maps=torch.tensor(maps), R=torch.tensor(R), T=torch.tensor(T))


@pytest.mark.parametrize("seed,F,K,img,sigma", [(0, 60, 8, 48, 1e-4), (1, 200, 50, 64, 1e-4), (2, 40, 3, (24, 40), 3e-3),
                                                (3, 500, 20, 32, 1e-4)])
def test_triangle_soup_matches_oracle(seed, F, K, img, sigma):
    """nothing sphere-like: overlapping, degenerate, huge, tiny and behind-the-camera faces, per-mesh face tables"""
    from unicorn_amd.renderer import Renderer
    from unicorn_amd.structures import Meshes, TexturesUV
    N = 2
    H, W = (img, img) if isinstance(img, int) else img
    sc = _soup(N, F, seed)
    dev = torch.device("cuda:0")
    rend = Renderer(img, cameras={"name": "fov", "fov": 30}, faces_per_pixel=K, sigma=sigma).to(dev)
    verts = sc["verts"].to(dev).requires_grad_(True)
    maps = sc["maps"].to(dev).requires_grad_(True)
    R, T = sc["R"].to(dev).requires_grad_(True), sc["T"].to(dev).requires_grad_(True)
    fb = sc["faces"].to(dev)[None].expand(N, -1, -1)
    ub = sc["verts_uvs"].to(dev)[None].expand(N, -1, -1)
    cot = np.random.RandomState(seed).randn(N, 4, H, W).astype(np.float32)
    ref, (gv, gR, gT, gm) = _oracle(sc, H, W, K, S30, sigma, cot)
    outs = []
    for fused in (True, False):
        rend.renderer.fused = fused
        for t in (verts, maps, R, T):
            t.grad = None
        out = rend(Meshes(verts=verts, faces=fb, textures=TexturesUV(maps, fb, ub)), R, T)
        out.backward(torch.tensor(cot, device=dev))
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref, atol=1e-5, rtol=0)

        def oracle64():
            g = _oracle(sc, H, W, K, S30, sigma, cot, dtype=np.float64)[1]
            return dict(verts=g[0], R=g[1], T=g[2], maps=g[3])
        assert_grads_to_the_bar({k: t.grad.cpu().numpy() for k, t in (("verts", verts), ("maps", maps), ("R", R), ("T", T))},
                                dict(verts=gv, R=gR, T=gT, maps=gm), oracle64, f"soup seed {seed} fused={fused}")
        outs.append(out.detach())
    assert torch.equal(outs[0], outs[1])


def test_debug_returns_blend_intermediates():
    """debug=True (renderer.py:199-200): (colors + background layer, alpha + trailing 1, occ_alpha, image)"""
    sc, rend, meshes, (verts, maps, R, T) = _setup(2, 2, 32, 6, seed=8, debug=True)
    sc2, plain, meshes2, leaves2 = _setup(2, 2, 32, 6, seed=8)
    colors, alpha, occ, image = rend(meshes, R, T)
    assert colors.shape == (2, 32, 32, 7, 3) and alpha.shape == occ.shape == (2, 32, 32, 7)
    assert torch.equal(image, plain(meshes2, leaves2[2], leaves2[3]))
    # the reference's own composition of the intermediates (renderer.py:196-197) gives the image back
    rgb = (occ[..., None] * alpha[..., None] * colors).sum(-2).permute(0, 3, 1, 2).detach()
    assert float((rgb - image.detach()[:, :3]).abs().max()) <= 1e-5
    assert float(((1 - occ.detach()[..., -1]) - image.detach()[:, 3]).abs().max()) <= 1e-6
    assert torch.all(colors[..., -1, :] == 1) and torch.all(alpha[..., -1] == 1) and torch.all(occ[..., 0] == 1)
    image.sum().backward()
    assert verts.grad is not None and maps.grad is not None


def test_one_call_render_step_equals_the_composed_entry_points():
    """ur_render_meshes_forward/backward (what Renderer.forward uses) against ur_project_faces + ur_render_layered_*
    (+ ur_project_faces_backward), the route of batches of unequal meshes: same bits forward, same gradients"""
    from unicorn_amd import ops
    N, H, K = 3, 48, 12
    sc, rend, meshes, (verts, maps, R, T) = _setup(N, 3, H, K, seed=31)
    faces = sc["faces"].cuda()
    fuv, shared = meshes.textures.faces_verts_uvs()
    first, nf = rend.renderer.rasterizer._mesh_index(N, faces.shape[0], verts.device)
    blur = orc.blur_radius_for(1e-4)
    cot = torch.randn(N, 4, H, H, device=verts.device, generator=torch.Generator(device=verts.device).manual_seed(3))
    one = ops.render_meshes(verts, maps, R, T, faces, S30, S30, first, nf, (H, H), blur, K, True, fuv, shared, 1e-4,
                            (1.0, 1.0, 1.0), None, 1)
    one.backward(cot)
    g_one = [t.grad.clone() for t in (verts, maps, R, T)]
    for t in (verts, maps, R, T):
        t.grad = None
    fv = ops.project_faces(verts, faces, R, T, S30, S30)
    two = ops.render_layered(fv, maps, first, nf, (H, H), blur, K, True, fuv, shared, 1e-4, (1.0, 1.0, 1.0), None, 1)
    two.backward(cot)
    assert torch.equal(one.detach(), two.detach())
    for a, t in zip(g_one, (verts, maps, R, T)):
        assert rel_err(a.cpu().numpy(), t.grad.cpu().numpy()) <= 1e-5
    # a second backward through the same graph (the forward's pre-cleared accumulators are used once only)
    img = ops.render_meshes(verts, maps, R, T, faces, S30, S30, first, nf, (H, H), blur, K, True, fuv, shared, 1e-4,
                            (1.0, 1.0, 1.0), None, 1)
    for t in (verts, maps, R, T):
        t.grad = None
    img.backward(cot, retain_graph=True)
    first_pass = verts.grad.clone()
    verts.grad = None
    img.backward(cot)
    assert rel_err(verts.grad.cpu().numpy(), first_pass.cpu().numpy()) <= 1e-5


def test_meshes_built_the_reference_way_render_like_plain_tensors():
    """unicorn.py:259-266 builds the batch as template.extend(B) -> offset_verts_(packed displacements) -> textures =
    TexturesUV(maps, faces[None].expand, uvs.expand) -> scale_verts_((B, 3) scales), and renders it (unicorn.py:233):
    same picture and same gradients as a batch built from the final vertex tensor directly"""
    from unicorn_amd.renderer import Renderer
    from unicorn_amd.structures import Meshes, TexturesUV
    dev = torch.device("cuda:0")
    B, H = 3, 48
    sc = synth.make_scene(B, level=2, tex_size=16, seed=9, device=dev)
    tmpl_v, faces = sc["verts"][0] * 0 + synth.icosphere(2)[0].to(dev) * 0.5, sc["faces"]
    uvs = sc["verts_uvs"][None]
    g = torch.Generator(device=dev).manual_seed(1)
    disp0 = 0.03 * torch.randn(B * tmpl_v.shape[0], 3, device=dev, generator=g)
    scales0 = 1.0 + 0.2 * torch.rand(B, 3, device=dev, generator=g)
    maps0 = sc["maps"].permute(0, 3, 1, 2).contiguous()                    # BCHW, as the texture generator emits them
    cot = torch.randn(B, 4, H, H, device=dev, generator=g)
    rend = Renderer(H, cameras={"name": "fov", "fov": 30}, faces_per_pixel=10, sigma=1e-4).to(dev)

    def leaves():
        return [t.clone().requires_grad_(True) for t in (disp0, scales0, maps0)]
    disp, scales, maps = leaves()
    meshes = Meshes(tmpl_v[None], faces[None]).extend(B)
    meshes.offset_verts_(disp)
    meshes.textures = TexturesUV(maps.permute(0, 2, 3, 1), faces[None].expand(B, -1, -1), uvs.expand(B, -1, -1))
    meshes.scale_verts_(scales)
    img = rend(meshes, sc["R"], sc["T"])
    img.backward(cot)
    d2, s2, m2 = leaves()
    verts = (tmpl_v[None] + d2.view(B, -1, 3)) * s2[:, None]
    fb = faces[None].expand(B, -1, -1)
    ref = rend(Meshes(verts, fb, TexturesUV(m2.permute(0, 2, 3, 1), fb, uvs.expand(B, -1, -1))), sc["R"], sc["T"])
    ref.backward(cot)
    assert torch.equal(img, ref) and 0.03 < float(img[:, 3].detach().mean()) < 0.9
    for a, b in ((disp, d2), (scales, s2), (maps, m2)):
        assert rel_err(a.grad.cpu().numpy(), b.grad.cpu().numpy()) <= 2e-5


def test_empty_batch_renders_to_an_empty_image_tensor():
    from unicorn_amd.renderer import Renderer
    from unicorn_amd.structures import Meshes, TexturesUV
    dev = torch.device("cuda:0")
    sc = synth.make_scene(1, level=1, tex_size=8, seed=0, device=dev)
    fb, ub = sc["faces"][None].expand(1, -1, -1), sc["verts_uvs"][None].expand(1, -1, -1)
    rend = Renderer((24, 40), faces_per_pixel=5, viz_size=(16, 16)).to(dev)
    none = Meshes(sc["verts"][:0], fb[:0], TexturesUV(sc["maps"][:0], fb[:0], ub[:0]))
    assert rend(none, sc["R"][:0], sc["T"][:0]).shape == (0, 4, 24, 40)
    assert rend(none, sc["R"][:0], sc["T"][:0], viz_purpose=True).shape == (0, 4, 16, 16)


def test_neighbour_swap_batches_reuse_tables_and_textures_of_the_rendered_batch():
    """unicorn.py:350,389,407: the neighbour loss takes verts_padded(), faces_padded() and .textures of the batch it has
    just rendered, and builds two more batches from them - swapped shapes `Meshes((verts / scales) * sh_S, faces,
    sh_tx)` and swapped textures `Meshes(tx_verts * tx_S, faces, textures)`.  Both render like batches built from plain
    tensors, and the gradients reach the vertices and the texture maps of the first batch through the shared objects"""
    from unicorn_amd.renderer import Renderer
    from unicorn_amd.structures import Meshes, TexturesUV
    dev = torch.device("cuda:0")
    B, H = 3, 40
    sc = synth.make_scene(B, level=2, tex_size=16, seed=4, device=dev)
    faces, uvs = sc["faces"], sc["verts_uvs"][None]
    fb, ub = faces[None].expand(B, -1, -1), uvs.expand(B, -1, -1)
    g = torch.Generator(device=dev).manual_seed(2)
    tmpl = synth.icosphere(2)[0].to(dev) * 0.5
    scales = (1.0 + 0.2 * torch.rand(B, 3, device=dev, generator=g))[:, None]
    sh_S, tx_S = [(1.0 + 0.2 * torch.rand(B, 1, 3, device=dev, generator=g)) for _ in range(2)]
    tx_verts = tmpl[None] + 0.03 * torch.randn(B, tmpl.shape[0], 3, device=dev, generator=g)
    sh_maps = torch.rand(B, 16, 16, 3, device=dev, generator=g)
    cots = [torch.randn(B, 4, H, H, device=dev, generator=g) for _ in range(3)]
    rend = Renderer(H, cameras={"name": "fov", "fov": 30}, faces_per_pixel=10, sigma=1e-4).to(dev)

    def run(shared):
        disp = (0.03 * torch.ones(B * tmpl.shape[0], 3, device=dev)).requires_grad_(True)
        maps = sc["maps"].clone().requires_grad_(True)
        meshes = Meshes(tmpl[None], faces[None]).extend(B)
        meshes.offset_verts_(disp)
        meshes.textures = TexturesUV(maps, fb, ub)
        meshes.scale_verts_(scales[:, 0])
        if shared:
            verts, f, textures = meshes.verts_padded(), meshes.faces_padded(), meshes.textures
            sh_tx = TexturesUV(sh_maps, f, ub)
        else:
            verts = (tmpl[None] + disp.view(B, -1, 3)) * scales
            f, textures, sh_tx = fb.contiguous(), TexturesUV(maps, fb.contiguous(), ub.contiguous()), TexturesUV(sh_maps, fb, ub)
            meshes = Meshes(verts, f, textures)
        batches = [meshes, Meshes((verts / scales) * sh_S, f, sh_tx), Meshes(tx_verts * tx_S, f, textures)]
        imgs = [rend(m, sc["R"], sc["T"]) for m in batches]
        sum((i * c).sum() for i, c in zip(imgs, cots)).backward()
        return imgs, disp.grad, maps.grad
    imgs_a, gd_a, gm_a = run(True)
    imgs_b, gd_b, gm_b = run(False)
    for a, b in zip(imgs_a, imgs_b):
        assert torch.equal(a, b) and 0.03 < float(a[:, 3].detach().mean()) < 0.9
    assert rel_err(gd_a.cpu().numpy(), gd_b.cpu().numpy()) <= 2e-5
    assert rel_err(gm_a.cpu().numpy(), gm_b.cpu().numpy()) <= 2e-5
    assert float(gm_a.abs().sum()) > 0 and float(gd_a.abs().sum()) > 0


def test_duplicate_faces_and_a_mesh_behind_the_camera():
    """exact depth ties through the fused path (three copies of a face: the smaller face index comes first, SURVEY 8a
    spec block) and a mesh that lies wholly behind the camera (background only, zero gradients)"""
    from unicorn_amd.renderer import Renderer
    from unicorn_amd.structures import Meshes, TexturesUV
    N, F, K, H = 2, 40, 6, 40
    sc = _soup(N, F, seed=11, behind=False)
    v = sc["verts"].view(N, F, 3, 3)
    v[0, 6] = v[0, 5]
    v[0, 7] = v[0, 5]
    v[0, 5:8, :, :2] *= 0.0
    v[0, 5:8, :, :2] += torch.tensor([[-0.3, -0.3], [0.35, -0.25], [0.0, 0.4]])      # a large face in view, three times over
    v[0, 5:8, :, 2] = 2.0
    v[1, :, :, 2] = -v[1, :, :, 2].abs() - 0.5                                        # mesh 1: everything behind the camera
    dev = torch.device("cuda:0")
    rend = Renderer(H, cameras={"name": "fov", "fov": 30}, faces_per_pixel=K, sigma=1e-4).to(dev)
    verts = sc["verts"].to(dev).requires_grad_(True)
    maps = sc["maps"].to(dev).requires_grad_(True)
    R, T = sc["R"].to(dev).requires_grad_(True), sc["T"].to(dev).requires_grad_(True)
    fb = sc["faces"].to(dev)[None].expand(N, -1, -1)
    ub = sc["verts_uvs"].to(dev)[None].expand(N, -1, -1)
    cot = np.random.RandomState(5).randn(N, 4, H, H).astype(np.float32)
    ref, (gv, gR, gT, gm) = _oracle(sc, H, H, K, S30, 1e-4, cot)
    meshes = Meshes(verts=verts, faces=fb, textures=TexturesUV(maps, fb, ub))
    out = rend(meshes, R, T)
    out.backward(torch.tensor(cot, device=dev))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, atol=1e-5, rtol=0)
    assert torch.equal(out[1, :3].detach(), torch.ones_like(out[1, :3])) and float(out[1, 3].detach().abs().max()) == 0.0
    assert float(verts.grad[1].abs().max()) == 0.0 and float(maps.grad[1].abs().max()) == 0.0
    assert float(R.grad[1].abs().max()) == 0.0 and float(T.grad[1].abs().max()) == 0.0

    def oracle64():
        g = _oracle(sc, H, H, K, S30, 1e-4, cot, dtype=np.float64)[1]
        return dict(verts=g[0], R=g[1], T=g[2], maps=g[3])
    assert_grads_to_the_bar({k: t.grad.cpu().numpy() for k, t in (("verts", verts), ("maps", maps), ("R", R), ("T", T))},
                            dict(verts=gv, R=gR, T=gT, maps=gm), oracle64, "duplicates")
    # the fragments of the same scene: the three copies occupy consecutive layers in face-index order wherever they show
    frag = rend.renderer.rasterizer(meshes, R=R, T=T)
    p2f = frag.pix_to_face[0].cpu().numpy()
    where5 = np.argwhere(p2f == 5)
    assert len(where5) > 50
    for y, x, k in where5[:200]:
        assert k + 2 < K and p2f[y, x, k + 1] == 6 and p2f[y, x, k + 2] == 7
