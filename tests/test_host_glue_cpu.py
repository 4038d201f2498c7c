"""Host-side caches of the eager path (no GPU): they must never serve stale data."""
import torch

from unicorn_amd import native
from unicorn_amd.renderer import MeshRasterizer, _rgb3


def test_shared_faces_table_follows_the_storage():
    rast = MeshRasterizer()
    base = torch.arange(12, dtype=torch.int64).view(4, 3)
    batch = base[None].expand(5, -1, -1)                       # unicorn.py:86-93 builds the batch like this
    t0 = rast._shared_faces(batch)
    assert t0.shape == (4, 3) and torch.equal(t0, base)
    assert rast._shared_faces(base[None].expand(5, -1, -1)) is t0   # a new view of the same storage: the kept table
    base[0, 0] = 99                                            # in-place edit: version counter moves, entry is not reused
    t1 = rast._shared_faces(batch)
    assert t1[0, 0].item() == 99
    other = torch.zeros(4, 3, dtype=torch.int64)[None].expand(5, -1, -1)
    assert torch.equal(rast._shared_faces(other), torch.zeros(4, 3, dtype=torch.int64))
    full = torch.arange(5 * 4 * 3, dtype=torch.int64).view(5, 4, 3)     # per-mesh topologies: handed through
    out = rast._shared_faces(full)
    assert out.shape == (5, 4, 3) and out.is_contiguous() and torch.equal(out, full)
    for _ in range(20):                                        # the cache stays small
        rast._shared_faces(torch.zeros(2, 3, dtype=torch.int64)[None].expand(3, -1, -1))
    assert len(rast._faces_cache) <= 9


def test_rgb3_forms():
    assert _rgb3((1, 1, 1)) == (1.0, 1.0, 1.0)
    assert _rgb3(((0.5, 0.25, 1.0),)) == (0.5, 0.25, 1.0)
    assert _rgb3(((0.5, 0.25, 1.0),)) == (0.5, 0.25, 1.0)      # second time: from the memo
    assert _rgb3([0.1, 0.2, 0.3]) == (0.1, 0.2, 0.3)            # lists are not hashable: computed each time
    t = torch.tensor([[0.2, 0.4, 0.6]])
    assert _rgb3(t) == tuple(float(v) for v in t.flatten())
    t[0, 0] = 0.9                                              # tensors are never memoised
    assert abs(_rgb3(t)[0] - 0.9) < 1e-7
    assert _rgb3((t,))[0] == _rgb3(t)[0]                       # a tuple holding a tensor: no memo either


def test_step_layout_parts_are_aligned_and_disjoint(monkeypatch):
    sizes = {"ur_layers_elems": 1000003, "ur_workspace_bytes": 77777, "ur_backward_scratch_bytes": 4099}
    monkeypatch.setattr(native, "_sized", lambda name, *a: sizes[name])
    native._step_layouts.clear()
    o_lay, o_ws, ws_b, o_scr, scr_b, o_acc, total = native._step_layout(3, 7, 16, 16, 5)
    native._step_layouts.clear()
    NF = 21
    assert o_lay >= NF * 36 and o_ws >= o_lay + sizes["ur_layers_elems"] * 4
    assert ws_b == 77777 and o_scr >= o_ws + ws_b and scr_b == 4099
    assert o_acc >= o_scr + scr_b and total >= o_acc + NF * 36
    assert all(o % 256 == 0 for o in (o_lay, o_ws, o_scr, o_acc, total))


def test_colour_arrays_outlive_cache_eviction():
    """the arrays handed to a C call must not depend on the colour cache keeping them"""
    native._rgb_cache.clear()
    first = native._rgb((0.25, 0.5, 0.75))
    addr = native._rgb_ptr(first)
    for i in range(600):                                       # far more distinct colours than the cache keeps
        native._rgb((float(i), 0.0, 1.0))
    assert len(native._rgb_cache) <= 257
    assert native._rgb_ptr(first) == addr and list(first) == [0.25, 0.5, 0.75]   # still alive: the caller holds it
    assert native._rgb_ptr(None) is None


def test_renderer_takes_options_like_the_reference():
    """renderer.py:26-57 reads its options with kwargs.get(name, default): unknown keys are ignored for construction
    and travel on in init_kwargs (configs of forks, model_kwargs of checkpoints)"""
    from unicorn_amd.renderer import Renderer
    kw = {"cameras": {"name": "fov", "fov": 30}, "faces_per_pixel": 7, "some_fork_option": {"a": 1}}
    r = Renderer(32, **kw)
    assert r.init_kwargs == kw and r.init_kwargs is not r.init_kwargs
    assert r.renderer.rasterizer.raster_settings.faces_per_pixel == 7
    r2 = Renderer((32, 48), **r.init_kwargs)
    assert r2.renderer.rasterizer.raster_settings.image_size == (32, 48)
    r.init_lights(name="directional", direction=((0, 0, 1),))   # renderer.py:65-74
    assert r.lights.direction == ((0, 0, 1),)
    r.update_lights(direction=((1, 0, 0),))
    r.reset_default_lights()
    assert r.lights.direction == ((0, 0, 1),)
    r.init_cameras(name="perspective", focal_length=3.7)        # renderer.py:59-63
    assert type(r.cameras).__name__ == "PerspectiveCameras"
