"""Parity at BASELINE.json's FULL sizes.  The oracle is too slow for whole batches there, so whole batches are
checked through size-independent properties of the path -- determinism, batch independence (an image depends on its
own mesh only), fused == two-stage bit for bit, linearity of the backward in the cotangent -- and a few images of each
batch are checked against the oracle one by one (forward and backward: the gradients of mesh n depend on
cotangent n only)."""
import numpy as np
import pytest
import torch

from oracle import orc
from unicorn_amd import synth

pytestmark = pytest.mark.gpu

S30 = float(1.0 / np.tan(np.deg2rad(15.0)))
DEV = "cuda:0"


def rel_err(a, b):
    return float(np.abs(a - b).max() / max(float(np.abs(b).max()), 1e-20))


class Batch:
    def __init__(self, N, level, img, K, seed, tex=64):
        from unicorn_amd.renderer import Renderer
        self.N, self.H, self.K = N, img, K
        self.sc = synth.make_scene(N, level=level, tex_size=tex, seed=seed)
        self.rend = Renderer(img, cameras={"name": "fov", "fov": 30}, faces_per_pixel=K, sigma=1e-4).to(DEV)
        self.leaves = [self.sc[k].to(DEV).requires_grad_(True) for k in ("verts", "maps", "R", "T")]
        self.faces, self.uvs = self.sc["faces"].to(DEV), self.sc["verts_uvs"].to(DEV)

    def meshes(self, sl=slice(None)):
        from unicorn_amd.structures import Meshes, TexturesUV
        v, m = self.leaves[0][sl], self.leaves[1][sl]
        n = v.shape[0]
        fb, ub = self.faces[None].expand(n, -1, -1), self.uvs[None].expand(n, -1, -1)
        return Meshes(verts=v, faces=fb, textures=TexturesUV(m, fb, ub))

    def render(self, sl=slice(None), fused=True):
        self.rend.renderer.fused = fused
        return self.rend(self.meshes(sl), self.leaves[2][sl], self.leaves[3][sl])

    def grads(self, cot, fused=True):
        for t in self.leaves:
            t.grad = None
        self.render(fused=fused).backward(cot)
        return [t.grad.detach().clone() for t in self.leaves]

    def oracle(self, n, cot):
        sc = self.sc
        return orc.render_fwd_bwd(sc["verts"][n:n + 1].numpy(), sc["faces"].numpy(), sc["R"][n:n + 1].numpy(),
                                  sc["T"][n:n + 1].numpy(), sc["maps"][n:n + 1].numpy(), sc["verts_uvs"][None].numpy(),
                                  self.H, self.H, self.K, S30, grad_image=cot[n:n + 1])

    def check_against_oracle(self, out, grads, cot, picks):
        for n in picks:
            ref, (gv, gR, gT, gm) = self.oracle(n, cot)
            np.testing.assert_allclose(out[n:n + 1].cpu().numpy(), ref, atol=1e-5, rtol=0)
            errs = [rel_err(got[n:n + 1].cpu().numpy(), want) for got, want in zip(grads, (gv, gm, gR, gT))]
            assert max(errs) <= 1e-4, (n, errs)


def test_headline_batch_b32_128px_k50():
    """the bench workload (BASELINE.json metric): B=32 level-3 icospheres, 128x128, K=50"""
    b = Batch(32, 3, 128, 50, seed=0)
    out = b.render().detach()
    assert out.shape == (32, 4, 128, 128)
    assert torch.equal(out, b.render().detach())                    # deterministic forward
    assert torch.equal(out, b.render(fused=False).detach())         # fused == two-stage, bit for bit
    assert torch.equal(out[5:9], b.render(slice(5, 9)).detach())    # an image depends on its own mesh only
    assert torch.equal(out[31:], b.render(slice(31, 32)).detach())
    a = out[:, 3]
    assert float(a.min()) >= 0.0 and float(a.max()) <= 1.0 and 0.05 < float(a.mean()) < 0.9
    assert bool(torch.all(out[:, :3].permute(0, 2, 3, 1)[a == 0] == 1.0))  # untouched pixels: white background
    rs = np.random.RandomState(0)
    c1 = rs.randn(*out.shape).astype(np.float32)
    c2 = rs.randn(*out.shape).astype(np.float32)
    g1 = b.grads(torch.from_numpy(c1).to(DEV))
    g2 = b.grads(torch.from_numpy(c2).to(DEV))
    g12 = b.grads(torch.from_numpy(c1 + 2.0 * c2).to(DEV))
    # the backward is linear in the cotangent -- up to fp32 rounding of per-pixel terms that are ~1/sigma larger than
    # their sums (the oracle comparison below is tighter because both sides round in the same order)
    lin = [rel_err((x + 2.0 * y).cpu().numpy(), z.cpu().numpy()) for x, y, z in zip(g1, g2, g12)]
    assert max(lin) <= 2e-3, lin
    g1_two = b.grads(torch.from_numpy(c1).to(DEV), fused=False)
    two = [rel_err(x.cpu().numpy(), y.cpu().numpy()) for x, y in zip(g1, g1_two)]
    assert max(two) <= 1e-4, two
    b.check_against_oracle(out, g1, c1, picks=(0, 17, 31))


def test_four_and_eight_wave_bins_agree():
    """The fused kernels run a bin with four waves of 64 pixels, or -- launches too small to fill the chip that way,
    N x bins <= 1024 -- with eight waves of 32 pixels, no bin schedule and an eight-wave backward.  Same pictures bit
    for bit, same gradients to rounding: a batch of 80 images (64x64: 1280 bins, four-wave) against its own slices of
    16 (256 bins, eight-wave), and against the oracle."""
    b = Batch(80, 2, 64, 50, seed=6, tex=32)
    out = b.render().detach()
    cot = np.random.RandomState(2).randn(*out.shape).astype(np.float32)
    g_all = b.grads(torch.from_numpy(cot).to(DEV))
    for lo in (0, 48, 64):
        sl = slice(lo, lo + 16)
        assert torch.equal(out[sl], b.render(sl).detach())
        for t in b.leaves:
            t.grad = None
        b.render(sl).backward(torch.from_numpy(cot[sl]).to(DEV))
        for full, leaf in zip(g_all, b.leaves):
            assert rel_err(leaf.grad[sl].cpu().numpy(), full[sl].cpu().numpy()) <= 2e-5
            assert float(leaf.grad[:lo].abs().max() if lo else 0.0) == 0.0   # nothing leaks into other meshes
    b.check_against_oracle(out, g_all, cot, picks=(3, 79))


def test_pose_step_batch_b192_64px_k50():
    """cfg2 pose step: 32 meshes x 6 pose candidates tiled K-major (utils/mesh.py:43-71, unicorn.py:232-233), 64x64"""
    from unicorn_amd.structures import repeat
    B, KP = 32, 6
    b = Batch(B, 3, 64, 50, seed=4)
    poses = synth.make_scene(B * KP, level=0, tex_size=2, seed=9)
    R6, T6 = poses["R"].to(DEV).requires_grad_(True), poses["T"].to(DEV).requires_grad_(True)
    out = b.rend(repeat(b.meshes(), KP), R6, T6)
    assert out.shape == (B * KP, 4, 64, 64)
    for k in (0, 3, 5):
        one = b.rend(b.meshes(), R6[k * B:(k + 1) * B], T6[k * B:(k + 1) * B])
        assert torch.equal(out[k * B:(k + 1) * B].detach(), one.detach())
    cot = np.random.RandomState(1).randn(*out.shape).astype(np.float32)
    out.backward(torch.from_numpy(cot).to(DEV))
    verts, maps = b.leaves[0], b.leaves[1]
    sc = b.sc
    for n in (0, 100, 191):   # candidate n renders mesh n % B under pose n
        m = n % B
        ref, (gv, gR, gT, gm) = orc.render_fwd_bwd(
            sc["verts"][m:m + 1].numpy(), sc["faces"].numpy(), poses["R"][n:n + 1].numpy(), poses["T"][n:n + 1].numpy(),
            sc["maps"][m:m + 1].numpy(), sc["verts_uvs"][None].numpy(), 64, 64, 50, S30, grad_image=cot[n:n + 1])
        np.testing.assert_allclose(out[n:n + 1].detach().cpu().numpy(), ref, atol=1e-5, rtol=0)
        errs = [rel_err(R6.grad[n:n + 1].cpu().numpy(), gR), rel_err(T6.grad[n:n + 1].cpu().numpy(), gT)]
        assert max(errs) <= 1e-4, (n, errs)
    # a mesh's gradient is the sum over its six candidates
    m = 7
    gv_sum, gm_sum = 0.0, 0.0
    for k in range(KP):
        n = k * B + m
        _, (gv, _, _, gm) = orc.render_fwd_bwd(
            sc["verts"][m:m + 1].numpy(), sc["faces"].numpy(), poses["R"][n:n + 1].numpy(), poses["T"][n:n + 1].numpy(),
            sc["maps"][m:m + 1].numpy(), sc["verts_uvs"][None].numpy(), 64, 64, 50, S30, grad_image=cot[n:n + 1])
        gv_sum, gm_sum = gv_sum + gv.astype(np.float64), gm_sum + gm.astype(np.float64)
    errs = [rel_err(verts.grad[m:m + 1].cpu().numpy(), gv_sum), rel_err(maps.grad[m:m + 1].cpu().numpy(), gm_sum)]
    assert max(errs) <= 1e-4, errs


def test_finer_mesh_batch_b64_128px_ico4():
    """configs[3]: finer mesh (level-4 icosphere, 5120 faces), 128x128, B=64"""
    b = Batch(64, 4, 128, 50, seed=2)
    out = b.render().detach()
    assert torch.equal(out, b.render(fused=False).detach())
    assert torch.equal(out[40:44], b.render(slice(40, 44)).detach())
    cot = np.random.RandomState(2).randn(*out.shape).astype(np.float32)
    g = b.grads(torch.from_numpy(cot).to(DEV))
    b.check_against_oracle(out, g, cot, picks=(0, 63))


def test_stress_config_256px_k100_ico5():
    """configs[4]: 20480-face meshes, 256x256, K=100 (ten scan rounds, deep K-lists, the two-workgroups-per-CU LDS
    budget); two images, one of them against the oracle (O(H W F): ~25 s of CPU)"""
    b = Batch(2, 5, 256, 100, seed=6)
    out = b.render().detach()
    assert torch.equal(out, b.render().detach())
    assert torch.equal(out, b.render(fused=False).detach())
    assert torch.equal(out[1:], b.render(slice(1, 2)).detach())
    cot = np.random.RandomState(3).randn(*out.shape).astype(np.float32)
    g = b.grads(torch.from_numpy(cot).to(DEV))
    b.check_against_oracle(out, g, cot, picks=(1,))


def test_stress_config_at_shard_size_b128():
    """configs[4] at the size one GPU of the node holds (1024 meshes over 8 GPUs): B=128 meshes of 20480 faces,
    256x256, K=100 -- ~27 GB of layer rows, ~7 GB of K-list spill, 2.6 M faces through the per-mesh prep kernel (80
    parts per mesh) and its face masks.  Whole batch: determinism and batch independence of image and gradients;
    one image against the oracle, forward and backward."""
    b = Batch(128, 5, 256, 100, seed=11)
    out = b.render().detach()
    assert out.shape == (128, 4, 256, 256)
    assert torch.equal(out, b.render().detach())
    assert torch.equal(out[100:102], b.render(slice(100, 102)).detach())
    assert torch.equal(out[100:102], b.render(slice(100, 102), fused=False).detach())  # fused == two-stage
    a = out[:, 3]
    assert float(a.min()) >= 0.0 and float(a.max()) <= 1.0 and 0.05 < float(a.mean()) < 0.9
    cot = np.random.RandomState(5).randn(*out.shape).astype(np.float32)
    g = b.grads(torch.from_numpy(cot).to(DEV))
    assert all(bool(torch.isfinite(x).all()) for x in g)
    # gradients of mesh n depend on image n only: the same mesh rendered on its own
    sl = slice(77, 78)
    b.rend.renderer.fused = True
    for t in b.leaves:
        t.grad = None
    b.render(sl).backward(torch.from_numpy(cot[sl]).to(DEV))
    alone = [t.grad[sl].detach().clone() for t in b.leaves]
    errs = [rel_err(x.cpu().numpy(), y[sl].cpu().numpy()) for x, y in zip(alone, g)]
    assert max(errs) <= 1e-4, errs
    b.check_against_oracle(out, g, cot, picks=(77,))
