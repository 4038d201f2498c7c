"""`render_rotated_views` (reference renderer.py:219-257, wrapped by save_mesh_as_gif): a turntable of one mesh through
the visualisation path -- exercises `Renderer.forward(viz_purpose=True)`, `Meshes.extend / to`, `look_at_view_transform`,
and, with eye_light, the rebuilt phong renderer whose directional light follows the camera (`update_lights` with one
direction per view)."""
import numpy as np
import pytest
import torch

from unicorn_amd import synth

pytestmark = pytest.mark.gpu


def _mesh(dev):
    from unicorn_amd.structures import Meshes, TexturesUV
    v, f = synth.icosphere(2)
    v = v * torch.tensor(synth.SCALE_ELLIPSE) * 0.5
    maps = torch.rand(1, 16, 16, 3, generator=torch.Generator().manual_seed(0))
    return Meshes(v[None].to(dev), f[None].to(dev), TexturesUV(maps.to(dev), f[None].to(dev), synth.spherical_uvs(v)[None].to(dev)))


def test_turntable_matches_direct_calls():
    from unicorn_amd.cameras import look_at_view_transform
    from unicorn_amd.renderer import Renderer, render_rotated_views
    dev = torch.device("cuda:0")
    mesh = _mesh(dev)
    rend = Renderer(64, cameras={"name": "fov", "fov": 30}, viz_size=(48, 48)).to(dev)
    views = render_rotated_views(mesh, img_size=48, n_views=13, elev=20, dist=2.7, renderer=rend, device=dev)
    assert views.shape == (13, 3, 48, 48) and views.device.type == "cpu"
    assert torch.isfinite(views).all() and 0.0 <= float(views.min()) and float(views.max()) <= 1.0
    azim = torch.linspace(-180, 180, 13)
    R = look_at_view_transform(dist=1, elev=20, azim=azim[10:13], device=dev)[0]   # the second call's views
    T = torch.tensor([[0.0, 0.0, 2.7]], device=dev).expand(3, -1)
    direct = rend(mesh.extend(3), R, T, viz_purpose=True).clamp(0, 1).cpu()
    assert torch.equal(views[10:], direct[:, :3])
    assert 0.05 < float(direct[:, 3].mean()) < 0.9                                   # the object is in view
    assert not torch.equal(views[0], views[3])                                      # and it turns
    assert torch.allclose(views[0], views[12], atol=1e-4)                           # -180 and +180 degrees
    # over a background: where the object is not, the background shows
    bkg = torch.rand(3, 24, 24, generator=torch.Generator().manual_seed(1))
    over = render_rotated_views(mesh, img_size=48, n_views=3, elev=20, dist=2.7, renderer=rend, bkg=bkg, device=dev)
    big = torch.nn.functional.interpolate(bkg[None], size=(48, 48), mode="bilinear", align_corners=False)[0]
    assert torch.allclose(over[0, :, 0, 0], big[:, 0, 0], atol=1e-6)


def test_eye_light_follows_the_camera():
    from unicorn_amd.renderer import Renderer, render_rotated_views
    dev = torch.device("cuda:0")
    mesh = _mesh(dev)
    rend = Renderer(64, cameras={"name": "fov", "fov": 30}, viz_size=(48, 48)).to(dev)
    plain = render_rotated_views(mesh, img_size=48, n_views=12, dist=2.7, renderer=rend, device=dev)
    lit = render_rotated_views(mesh, img_size=48, n_views=12, dist=2.7, renderer=rend, eye_light=True, device=dev)
    assert lit.shape == plain.shape and torch.isfinite(lit).all()
    inside = (plain < 0.999).any(1, keepdim=True).expand_as(plain)  # object pixels (white background elsewhere)
    assert torch.equal(lit[~inside], plain[~inside])
    # ambient 0.7 + diffuse 0.3 * cos: never brighter than the texture, darker towards the silhouette
    assert float((lit - plain)[inside].max()) <= 1e-5 and float((plain - lit)[inside].mean()) > 0.01
    # the headlight turns with the camera: the shading of a view equals the first view's, up to the object's own turn
    ratio = (lit[:, :, 20:28, 20:28].sum(1) / plain[:, :, 20:28, 20:28].sum(1).clamp_min(1e-3)).mean((1, 2))
    assert float(ratio.max() - ratio.min()) < 0.05 and float(ratio.mean()) > 0.9   # the centre faces the light in EVERY view
    with pytest.raises(NotImplementedError):
        render_rotated_views(mesh, img_size=48, n_views=2, renderer=rend, eye_light=True, R=torch.eye(3), device=dev)
