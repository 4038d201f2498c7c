"""ur_mesh_regularizers (normal consistency + uniform Laplacian, value and gradient in one launch) against the CPU
oracle and against the golden produced by the reference's own normal_consistency (utils/mesh.py:92-139)."""
import os

import numpy as np
import pytest
import torch

from oracle import orc
from unicorn_amd import synth

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.abs(a - b).max() / max(float(np.abs(b).max()), 1e-20))


def _meshes(verts, faces):
    from unicorn_amd.structures import Meshes
    N = verts.shape[0]
    return Meshes(verts=verts, faces=faces[None].expand(N, -1, -1))


def test_matches_reference_golden(golden_dir):
    from unicorn_amd import regularizers
    g = np.load(os.path.join(golden_dir, "regularizers.npz"))
    dev = "cuda:0"
    verts = torch.tensor(g["verts"], dtype=torch.float32, device=dev, requires_grad=True)
    m = _meshes(verts, torch.tensor(g["faces"], device=dev))
    normal, lap = regularizers.mesh_regularizers(m)
    assert abs(float(normal.detach()) - float(g["normal"])) < 1e-5 and abs(float(lap.detach()) - float(g["laplacian"])) < 1e-6
    (0.7 * normal + 0.3 * lap).backward()
    want = 0.7 * g["grad_normal"] + 0.3 * g["grad_laplacian"]
    assert rel(verts.grad.cpu().numpy(), want) < 1e-4
    # the two public names of the reference's call sites (unicorn.py:340-343)
    assert float(regularizers.normal_consistency(m).detach()) == float(normal.detach())
    assert float(regularizers.laplacian_smoothing(m, method="uniform").detach()) == float(lap.detach())


@pytest.mark.parametrize("N,level", [(8, 3), (2, 5)], ids=["ico3-lds", "ico5-global-atomics"])
def test_matches_oracle(N, level):
    from unicorn_amd import native, regularizers
    sc = synth.make_scene(N, level=level, tex_size=8, seed=2)
    quads, off, idx = regularizers.build_topology(sc["faces"].numpy())
    v = sc["verts"].numpy()
    dev = "cuda:0"
    t = lambda a: torch.tensor(a, device=dev)
    nsum, lsum, gN, gL = native.mesh_regularizers(t(v), t(quads), t(off), t(idx), True)
    rn, rl, rgN, rgL = orc.mesh_regularizers(v.astype(np.float64), quads, off, idx, dtype=np.float64)
    assert rel(nsum.cpu().numpy(), rn) < 1e-4 and rel(lsum.cpu().numpy(), rl) < 1e-5
    assert rel(gN.cpu().numpy(), rgN) < 2e-4 and rel(gL.cpu().numpy(), rgL) < 1e-4
    nsum2, lsum2, gN2, gL2 = native.mesh_regularizers(t(v), t(quads), t(off), t(idx), False)
    assert gN2 is None and gL2 is None and torch.equal(lsum2, lsum)


def test_rejects_open_meshes():
    from unicorn_amd import regularizers
    with pytest.raises(ValueError):
        regularizers.build_topology(np.array([[0, 1, 2], [0, 2, 3]]))


def test_the_reference_call_pattern_shares_one_launch():
    """unicorn.py:259-266 + 340-343: template.extend(B) (the copies share the template's face table), offset_verts_,
    then normal_consistency(meshes) and laplacian_smoothing(meshes) as two calls: the second is served from the first's
    launch; the terms can be back-propagated together or one after the other; a changed vertex tensor is seen"""
    from unicorn_amd import regularizers
    from unicorn_amd.structures import Meshes
    dev = "cuda:0"
    v, f = synth.icosphere(2)
    v, f = v.to(dev), f.to(dev)
    B = 3
    disp0 = 0.05 * torch.randn(B * v.shape[0], 3, device=dev, generator=torch.Generator(device=dev).manual_seed(0))

    def batch(disp):
        m = Meshes(v[None], f[None]).extend(B)
        assert m.faces_padded().stride(0) == 0          # one table, not B copies
        m.offset_verts_(disp)
        return m
    d1 = disp0.clone().requires_grad_(True)
    m = batch(d1)
    n1 = regularizers.normal_consistency(m)
    l1 = regularizers.laplacian_smoothing(m, method="uniform")
    assert n1.grad_fn is l1.grad_fn                     # both from ONE autograd node = one launch
    (0.3 * n1 + 0.7 * l1).backward()
    d2 = disp0.clone().requires_grad_(True)
    n2, l2 = regularizers.mesh_regularizers(Meshes(v[None] + d2.view(B, -1, 3), f[None].expand(B, -1, -1)))
    (0.3 * n2 + 0.7 * l2).backward()
    assert float(n1.detach()) == float(n2.detach()) and float(l1.detach()) == float(l2.detach())
    assert torch.equal(d1.grad, d2.grad)
    # one after the other
    d3 = disp0.clone().requires_grad_(True)
    m3 = batch(d3)
    (0.3 * regularizers.normal_consistency(m3)).backward()
    (0.7 * regularizers.laplacian_smoothing(m3)).backward()
    assert rel(d3.grad.cpu().numpy(), d1.grad.cpu().numpy()) < 1e-6
    # new vertices on the same batch object: not served from the old launch
    before = float(regularizers.normal_consistency(m3).detach())
    m3.scale_verts_(torch.tensor([1.0, 2.0, 0.5], device=dev).expand(B, 3))
    assert float(regularizers.normal_consistency(m3).detach()) != before
