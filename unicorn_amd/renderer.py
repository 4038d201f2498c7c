"""Drop-in for the reference's `Renderer` (src/model/renderer.py:25-116): same constructor kwargs, same
`forward(meshes, R, T, viz_purpose=False) -> (N,4,H,W)`, `to`, `compute_vertex_visibility`, `update_lights`,
`reset_default_lights`, `init_kwargs`, `.cameras`, `.lights`, `.renderer`, `.viz_renderer.rasterizer`.

Everything numeric runs in the hand-written HIP kernels behind include/unicorn_render.h; this file only composes them.
"""
import math
from copy import deepcopy
from dataclasses import dataclass, field

import torch
from torch import nn
from torch.nn import functional as F

from . import lighting, native, ops
from .cameras import FoVPerspectiveCameras, PerspectiveCameras, look_at_view_transform
from .structures import Meshes, TexturesUV, TexturesVertex  # noqa: F401  (re-exported for callers)

LAYERED_SHADER = True
SHADING_TYPE = 'raw'
VIZ_IMG_SIZE = 256


class BlendParams:
    def __init__(self, sigma=1e-4, gamma=1e-4, background_color=(1.0, 1.0, 1.0)):
        self.sigma, self.gamma, self.background_color = sigma, gamma, background_color


class RasterizationSettings:
    def __init__(self, image_size=256, blur_radius=0.0, faces_per_pixel=1, perspective_correct=False,
                 clip_barycentric_coords=None, cull_backfaces=False, **unused):
        self.image_size = (image_size, image_size) if isinstance(image_size, int) else tuple(image_size)
        self.blur_radius, self.faces_per_pixel = float(blur_radius), int(faces_per_pixel)
        self.perspective_correct, self.cull_backfaces = perspective_correct, cull_backfaces
        # pytorch3d: clip_barycentric_coords defaults to blur_radius > 0
        self.clip_barycentric_coords = (self.blur_radius > 0.0) if clip_barycentric_coords is None \
            else clip_barycentric_coords


class Fragments:
    """pix_to_face (packed indices, -1 = empty), zbuf, bary_coords, dists, all (N,H,W,K[,3]).  The kernels keep
    int32 indices; `.pix_to_face` widens to int64 on first use (what pytorch3d hands to callers, renderer.py:108)."""

    def __init__(self, pix_to_face_i32, zbuf, bary_coords, dists):
        self.pix_to_face_i32, self.zbuf, self.bary_coords, self.dists = pix_to_face_i32, zbuf, bary_coords, dists
        self._p2f64 = None

    @property
    def pix_to_face(self):
        if self._p2f64 is None:
            self._p2f64 = self.pix_to_face_i32.long()
        return self._p2f64


class AmbientLights:
    """Ambient-only lighting (reference: pytorch3d_monkey.py:7-26); colour (1,1,1) leaves textures untouched."""

    def __init__(self, ambient_color=((1.0, 1.0, 1.0),), device='cpu'):
        self.ambient_color = ambient_color
        self.device = device

    def to(self, device):
        self.device = device
        return self


class DirectionalLights(AmbientLights):
    def __init__(self, ambient_color=((0.5, 0.5, 0.5),), diffuse_color=((0.3, 0.3, 0.3),),
                 specular_color=((0.2, 0.2, 0.2),), direction=((0, 1, 0),), device='cpu'):
        super().__init__(ambient_color, device)
        self.diffuse_color, self.specular_color, self.direction = diffuse_color, specular_color, direction


class PointLights(AmbientLights):
    def __init__(self, ambient_color=((0.5, 0.5, 0.5),), diffuse_color=((0.3, 0.3, 0.3),),
                 specular_color=((0.2, 0.2, 0.2),), location=((0, 1, 0),), device='cpu'):
        super().__init__(ambient_color, device)
        self.diffuse_color, self.specular_color, self.location = diffuse_color, specular_color, location


def _f32(t):
    return t if t.dtype is torch.float32 else t.float()


_rgb3_seen = {}


def _rgb3(c):
    """colour given as tuple / nested list / tensor -> 3 python floats, without touching the GPU stream unless the
    caller stored a device tensor"""
    if type(c) is tuple:  # immutable and hashable (unless it holds tensors / lists): remembered
        try:
            hit = _rgb3_seen.get(c)
        except TypeError:
            hit = None
        if hit is not None:
            return hit
    if not torch.is_tensor(c):  # plain numbers (the configs' case): no tensor round trip
        flat = c
        while isinstance(flat, (tuple, list)) and len(flat) == 1:
            flat = flat[0]
        if isinstance(flat, (tuple, list)) and len(flat) == 3 and all(isinstance(v, (int, float)) for v in flat):
            out = (float(flat[0]), float(flat[1]), float(flat[2]))
            if type(c) is tuple and len(_rgb3_seen) < 64:
                try:
                    _rgb3_seen[c] = out
                except TypeError:
                    pass
            return out
        if torch.is_tensor(flat):  # a tensor wrapped in one-element tuples / lists
            c = flat
    t = torch.as_tensor(c, dtype=torch.float32).detach().cpu().flatten()
    assert t.numel() == 3, "colour must have 3 components"
    return tuple(float(v) for v in t)


class MeshRasterizer(nn.Module):
    """meshes + (R, T) -> Fragments.  Replaces pytorch3d MeshRasterizer (transform + rasterize_meshes)."""

    def __init__(self, cameras=None, raster_settings=None):
        super().__init__()
        self.cameras = cameras
        self.raster_settings = raster_settings if raster_settings is not None else RasterizationSettings()
        self._idx_cache = {}
        self._faces_cache = {}

    def to(self, device):
        if self.cameras is not None:
            self.cameras = self.cameras.to(device)
        return self

    def _mesh_index(self, N, F, device):
        key = (N, F, device)
        if key not in self._idx_cache:
            first = torch.arange(N, dtype=torch.int64, device=device) * F
            self._idx_cache[key] = (first, torch.full((N,), F, dtype=torch.int64, device=device))
        return self._idx_cache[key]

    def _shared_faces(self, faces):
        """(N,F,3) faces -> what the one-call path takes: the (F,3) table when the batch is an expanded view of one
        topology (unicorn.py:86-93 builds it so), else the contiguous (N,F,3).  The reference hands over the same
        expanded tensor every step: the table view is kept for as long as that storage is (no tensor op per step)."""
        if faces.stride(0) != 0:
            return faces.contiguous()
        key = (faces.data_ptr(), faces._version, faces.shape, faces.device)
        hit = self._faces_cache.get(key)
        if hit is None:
            if len(self._faces_cache) > 8:
                self._faces_cache.clear()
            # (the entry keeps `faces` alive, so its address cannot pass to another tensor meanwhile)
            hit = self._faces_cache[key] = (faces[0].contiguous(), faces)
        return hit[0]

    def transform(self, meshes, R, T):
        """-> face_verts (sum F_n, 3, 3) in NDC (z = view depth), mesh_first_face, mesh_num_faces"""
        sx, sy = self.cameras.focal_scale()
        if meshes.equisized:
            verts, faces = meshes.verts_padded(), meshes.faces_padded()
            N, Fn = faces.shape[:2]
            faces = faces[0] if faces.stride(0) == 0 else faces
            face_verts = ops.project_faces(verts.float(), faces.contiguous(), R.float(), T.float(), sx, sy)
            first, nf = self._mesh_index(N, Fn, verts.device)
            return face_verts, first, nf
        # heterogeneous batch: plain tensor ops on the packed representation (not the reference's case), in the
        # kernels' operation order -- ((p0 r0c + p1 r1c) + p2 r2c) + tc, then (s x) / z -- one rounding per torch op, so
        # a mesh renders bit for bit as it does alone through k_project_faces
        from .shading import gather_rows
        idx = meshes.verts_packed_to_mesh_idx()
        p = meshes.verts_packed().float()
        Ri, Ti = R.float()[idx], T.float()[idx]
        vx, vy, vz = (((p[:, 0] * Ri[:, 0, c] + p[:, 1] * Ri[:, 1, c]) + p[:, 2] * Ri[:, 2, c]) + Ti[:, c]
                      for c in range(3))
        ndc = torch.stack([(sx * vx) / vz, (sy * vy) / vz, vz], -1)
        return (gather_rows(ndc, meshes.faces_packed()), meshes.mesh_to_faces_packed_first_idx(),
                meshes.num_faces_per_mesh())

    def forward(self, meshes, R=None, T=None, **kwargs):
        rs = kwargs.get("raster_settings", self.raster_settings)
        face_verts, first, nf = self.transform(meshes, R, T)
        p2f, zbuf, bary, dists = ops.rasterize(face_verts, first, nf, rs.image_size, rs.blur_radius,
                                               rs.faces_per_pixel, rs.clip_barycentric_coords, rs.cull_backfaces)
        frag = Fragments(p2f, zbuf, bary, dists)
        frag.mesh_first_face = first
        return frag


class LayeredShader(nn.Module):
    """Reference: LayeredShader (renderer.py:127-172) + layered_rgb_blend (:175-202)."""

    def __init__(self, device='cpu', cameras=None, lights=None, materials=None, blend_params=None, clip_inside=True,
                 shading_type='raw', debug=False):
        super().__init__()
        self.lights = lights if lights is not None else AmbientLights()
        self.materials, self.cameras = materials, cameras
        self.blend_params = blend_params if blend_params is not None else BlendParams()
        self.clip_inside, self.shading_type, self.debug = clip_inside, shading_type, debug
        if shading_type not in ('raw', 'phong', 'flat', 'gouraud'):
            raise NotImplementedError
        if self.materials is None:
            self.materials = lighting.Materials()

    def to(self, device):
        if self.cameras is not None:
            self.cameras = self.cameras.to(device)
        self.lights = self.lights.to(device)
        return self

    def blend_mode(self, blend_params):
        if blend_params.sigma == 0:
            return native.BLEND_HARD
        return native.BLEND_CLIP_INSIDE if self.clip_inside else native.BLEND_SIGMOID

    def render_fused(self, rasterizer, meshes, R=None, T=None, **kwargs):
        bp = kwargs.get("blend_params", self.blend_params)
        rs = kwargs.get("raster_settings", rasterizer.raster_settings)
        n_images = len(meshes)
        one_call = meshes.equisized and not rs.cull_backfaces
        base = getattr(meshes, "tile_base", None)
        if base is not None and one_call and native.can_tile(base.verts_padded().shape[1],
                                                             base.faces_padded().shape[1], *rs.image_size):
            # utils.mesh.repeat(meshes, K) (unicorn.py:232-233) without the copies: the B meshes as they are, the K*B
            # poses index them (image n = mesh n % B); the gradients land in the B meshes' own tensors
            meshes = base
        tex = meshes.textures
        ambient = _rgb3(self.lights.ambient_color)
        ambient = None if ambient == (1.0, 1.0, 1.0) else ambient
        face_uvs, shared = tex.faces_verts_uvs()
        if one_call:
            # the reference's case (one template topology): camera + rasterizer + shader in one call each way
            sx, sy = rasterizer.cameras.focal_scale()
            verts, faces = meshes.verts_padded(), meshes.faces_padded()
            faces = rasterizer._shared_faces(faces)
            first, nf = rasterizer._mesh_index(n_images, faces.shape[-2], verts.device)
            return ops.render_meshes(_f32(verts), _f32(tex.maps_padded()), _f32(R), _f32(T), faces,
                                     sx, sy, first, nf, rs.image_size, rs.blur_radius, rs.faces_per_pixel,
                                     rs.clip_barycentric_coords, _f32(face_uvs), shared, bp.sigma,
                                     _rgb3(bp.background_color), ambient, self.blend_mode(bp))
        face_verts, first, nf = rasterizer.transform(meshes, R, T)
        return ops.render_layered(face_verts, tex.maps_padded().float(), first, nf, rs.image_size, rs.blur_radius,
                                  rs.faces_per_pixel, rs.clip_barycentric_coords, face_uvs.float(), shared, bp.sigma,
                                  _rgb3(bp.background_color), ambient, self.blend_mode(bp))

    def debug_outputs(self, colors, fragments, bp, image):
        """debug=True (renderer.py:199-200): the blend's intermediates next to the image -- colours with the
        background appended as layer K, the layers' alphas with a trailing 1, and the transmittance in front of every
        layer.  Inspection only; alphas and transmittances come from the library (ur_blend_intermediates: the blend
        kernels' own alpha function), the image from the blend kernel."""
        N, H, W, K = fragments.dists.shape
        alpha, occ_alpha = native.blend_intermediates(fragments.pix_to_face_i32, fragments.dists.detach().contiguous(),
                                                      bp.sigma, self.blend_mode(bp))
        bg = torch.tensor(_rgb3(bp.background_color), dtype=torch.float32, device=alpha.device)
        colors = torch.cat([colors, bg.expand(N, H, W, 1, 3)], dim=-2)
        return colors, alpha, occ_alpha, image

    def forward(self, fragments, meshes, **kwargs):
        bp = kwargs.get("blend_params", self.blend_params)
        tex = meshes.textures
        colors = None
        if self.shading_type != 'raw' or not isinstance(tex, TexturesUV):
            # lit colours (renderer.py:165-171) or vertex colours, then layered_rgb_blend over them (ur_blend_colors)
            colors = lit_colors(self.shading_type, meshes, fragments, kwargs.get("lights", self.lights),
                                kwargs.get("materials", self.materials), kwargs["R"], kwargs["T"])
            image = ops.blend_colors(colors, fragments.pix_to_face_i32, fragments.dists, None, self.blend_mode(bp),
                                     bp.sigma, bp.gamma, _rgb3(bp.background_color))
        else:
            ambient = _rgb3(self.lights.ambient_color)
            face_uvs, shared = tex.faces_verts_uvs()
            image = ops.shade_blend(fragments.pix_to_face_i32, fragments.bary_coords, fragments.dists,
                                    tex.maps_padded().float(), face_uvs.float(), shared, fragments.mesh_first_face,
                                    bp.sigma, _rgb3(bp.background_color),
                                    None if ambient == (1.0, 1.0, 1.0) else ambient, self.blend_mode(bp))
        if not self.debug:
            return image
        if colors is None:
            colors = lit_colors('raw', meshes, fragments, kwargs.get("lights", self.lights), None, None, None)
        return self.debug_outputs(colors, fragments, bp, image)


def layered_rgb_blend(colors, fragments, blend_params, clip_inside=True, debug=False):
    """The reference's blend function by its own name and signature (renderer.py:175-202): per-fragment colours
    (N,H,W,K,3) composited front to back over the background -> (N,4,H,W); with `debug` the (colours + background layer,
    alphas + trailing 1, transmittances, image) tuple of renderer.py:199-200.  One HIP kernel each way
    (ur_blend_colors / ur_blend_colors_backward); gradients flow to `colors` and `fragments.dists`."""
    shader = LayeredShader(blend_params=blend_params, clip_inside=clip_inside, shading_type='raw', debug=debug)
    image = ops.blend_colors(colors, fragments.pix_to_face_i32, fragments.dists, None, shader.blend_mode(blend_params),
                             blend_params.sigma, blend_params.gamma, _rgb3(blend_params.background_color))
    return shader.debug_outputs(colors, fragments, blend_params, image) if debug else image


def lit_colors(shading_type, meshes, fragments, lights, materials, R, T):
    """per-fragment colours (N,H,W,K,3) of LayeredShader's / SoftPhongShader's shading step"""
    if shading_type == 'gouraud':
        return lighting.gouraud_shading(meshes, fragments, lights, materials, R, T)
    texels = meshes.sample_textures(fragments)
    if shading_type == 'raw':
        amb = torch.as_tensor(lights.ambient_color, dtype=torch.float32, device=texels.device)
        return texels if bool(torch.all(amb == 1)) else texels * amb.view(-1, 1, 1, 1, 3)  # renderer.py:162-164
    fn = {'phong': lighting.phong_shading, 'flat': lighting.flat_shading}[shading_type]
    return fn(meshes, fragments, lights, materials, texels, R, T)


class SoftPhongShaderPlus(nn.Module):
    """Reference: SoftPhongShaderPlus (renderer.py:205-216) = pytorch3d SoftPhongShader (phong_shading +
    softmax_rgb_blend with the cameras' znear / zfar, defaults 1 / 100) with a BCHW output."""

    def __init__(self, device='cpu', cameras=None, lights=None, materials=None, blend_params=None, debug=False):
        super().__init__()
        self.lights = lights if lights is not None else PointLights()
        self.materials = materials if materials is not None else lighting.Materials()
        self.cameras = cameras
        self.blend_params = blend_params if blend_params is not None else BlendParams()

    def to(self, device):
        if self.cameras is not None:
            self.cameras = self.cameras.to(device)
        self.lights = self.lights.to(device)
        self.materials = self.materials.to(device)
        return self

    def forward(self, fragments, meshes, **kwargs):
        bp = kwargs.get("blend_params", self.blend_params)
        cams = kwargs.get("cameras", self.cameras)
        colors = lit_colors('phong', meshes, fragments, kwargs.get("lights", self.lights),
                            kwargs.get("materials", self.materials), kwargs["R"], kwargs["T"])
        znear = kwargs.get("znear", getattr(cams, "znear", 1.0))
        zfar = kwargs.get("zfar", getattr(cams, "zfar", 100.0))
        return ops.blend_colors(colors, fragments.pix_to_face_i32, fragments.dists, fragments.zbuf,
                                native.BLEND_SOFTMAX, bp.sigma, bp.gamma, _rgb3(bp.background_color), znear, zfar)


class MeshRenderer(nn.Module):
    """rasterizer + shader.  With a LayeredShader over UV textures (the training configuration) both stages and
    their backward run as ONE fused kernel each way (ur_render_layered_*); `fused=False` forces the two-stage path
    through materialised Fragments (same numbers, used by tests and by callers that want the fragments)."""

    def __init__(self, rasterizer, shader, fused=True):
        super().__init__()
        self.rasterizer, self.shader, self.fused = rasterizer, shader, fused

    def to(self, device):
        self.rasterizer.to(device)
        self.shader.to(device)
        return self

    def forward(self, meshes, **kwargs):
        shader = self.shader  # (a sub-module: every `self.shader` goes through nn.Module.__getattr__)
        if len(meshes) == 0:  # an empty batch (the last, ragged batch of a loader can be): nothing to launch
            H, W = kwargs.get("raster_settings", self.rasterizer.raster_settings).image_size
            return torch.zeros(0, 4, H, W, device=kwargs["R"].device)
        if self.fused and isinstance(shader, LayeredShader) and shader.shading_type == 'raw' \
                and isinstance(meshes.textures, TexturesUV) and not shader.debug:
            return shader.render_fused(self.rasterizer, meshes, **kwargs)
        fragments = self.rasterizer(meshes, **kwargs)
        return shader(fragments, meshes, **kwargs)


class VizMeshRenderer(MeshRenderer):
    """The visualisation pipeline (renderer.py:119-124): the scene is rasterized at twice the requested size and
    every 2x2 block of the result is averaged -- anti-aliasing by supersampling.  Never differentiated."""

    def forward(self, meshes, **kwargs):
        with torch.no_grad():
            supersampled = super().forward(meshes, **kwargs)
            return F.avg_pool2d(supersampled, 2)


def _pair(size):
    return (int(size), int(size)) if isinstance(size, int) else (int(size[0]), int(size[1]))


@dataclass(frozen=True)
class RenderConfig:
    """Resolved options of a `Renderer`.  Names and defaults are the reference's (renderer.py:26-57: `kwargs.get(name,
    default)`), because configs and checkpoints carry them (`model_kwargs`, unicorn.py:46-47)."""
    cameras: dict = field(default_factory=dict)     # {'name': 'fov' | 'perspective', + that camera's arguments}
    lights: dict = field(default_factory=dict)      # {'name': 'ambient' | 'directional' | 'point', + arguments}
    sigma: float = 1e-4
    background_color: tuple = (1, 1, 1)
    faces_per_pixel: int = 25
    layered_shader: bool = LAYERED_SHADER
    clip_inside: bool = True
    shading_type: str = SHADING_TYPE
    debug: bool = False
    viz_size: tuple = (VIZ_IMG_SIZE, VIZ_IMG_SIZE)


CAMERA_TYPES = {'fov': FoVPerspectiveCameras, 'perspective': PerspectiveCameras}
LIGHT_TYPES = {'ambient': AmbientLights, 'directional': DirectionalLights, 'point': PointLights}
# update_lights(direction, ka, kd, ks) -> attribute of the lights object (renderer.py:89-97)
LIGHT_FIELDS = (('direction', 'direction'), ('ka', 'ambient_color'), ('kd', 'diffuse_color'), ('ks', 'specular_color'))


def _from_spec(spec, types, default_name):
    """{'name': type name, **arguments} -> instance (renderer.py:59-74)"""
    args = {k: deepcopy(v) for k, v in spec.items() if k != 'name'}
    return types[spec.get('name', default_name)](**args)


class Renderer(nn.Module):
    """Drop-in for the reference's `Renderer` (renderer.py:25-116).  Two pipelines share one camera and one set of
    lights: `renderer` (soft rasterization for training: K faces per pixel, blur radius ln(1/1e-4 - 1) * sigma) and
    `viz_renderer` (hard, one face per pixel, 2x supersampled).  Options are read like the reference reads them
    (`kwargs.get(name, default)`, renderer.py:30-52): a key the class does not know is ignored for construction and
    kept in `init_kwargs`, which returns exactly what the caller passed (configs of forks and `model_kwargs` of
    checkpoints travel through it, unicorn.py:46-47)."""

    def __init__(self, img_size, **kwargs):
        super().__init__()
        self._init_kwargs = deepcopy(kwargs)
        known = {f for f in RenderConfig.__dataclass_fields__}
        self.config = cfg = RenderConfig(**deepcopy({k: v for k, v in kwargs.items() if k in known}))
        self.init_cameras(**cfg.cameras)
        self.init_lights(**cfg.lights)
        viz = _pair(cfg.viz_size)
        self.renderer = self._pipeline(MeshRenderer, _pair(img_size), cfg.faces_per_pixel, cfg.sigma)
        self.viz_renderer = self._pipeline(VizMeshRenderer, (2 * viz[0], 2 * viz[1]), 1, 0.0)

    def init_cameras(self, **kwargs):
        """renderer.py:59-63"""
        self.cameras = _from_spec(kwargs, CAMERA_TYPES, 'fov')

    def init_lights(self, **kwargs):
        """renderer.py:65-74; what reset_default_lights() goes back to after update_lights() (the eye-light GIFs,
        renderer.py:225-251) is remembered for every kind of light, not only the directional one"""
        self.lights = _from_spec(kwargs, LIGHT_TYPES, 'ambient')
        self._light_defaults = {attr: getattr(self.lights, attr) for _, attr in LIGHT_FIELDS if hasattr(self.lights, attr)}

    def _pipeline(self, kind, image_size, faces_per_pixel, sigma):
        """rasterizer + shader for one (size, K, sigma): renderer.py:45-49 (training) and :51-57 (visualisation).  The
        blur radius is where the blend's sigmoid falls to 1e-4: ln(1/1e-4 - 1) * sigma; 0 makes the raster hard."""
        cfg = self.config
        blend = BlendParams(sigma=sigma, background_color=cfg.background_color)
        raster = RasterizationSettings(image_size=image_size, blur_radius=math.log(1.0 / 1e-4 - 1.0) * sigma,
                                       faces_per_pixel=faces_per_pixel, perspective_correct=False)
        common = dict(cameras=self.cameras, lights=self.lights, blend_params=blend, debug=cfg.debug)
        if cfg.layered_shader:
            shader = LayeredShader(clip_inside=cfg.clip_inside, shading_type=cfg.shading_type, **common)
        else:
            shader = SoftPhongShaderPlus(**common)
        return kind(MeshRasterizer(cameras=self.cameras, raster_settings=raster), shader)

    @property
    def init_kwargs(self):
        return deepcopy(self._init_kwargs)

    def forward(self, meshes, R, T, viz_purpose=False):
        """-> (N, 4, H, W) RGBA"""
        pipeline = self.viz_renderer if viz_purpose else self.renderer
        return pipeline(meshes, R=R, T=T)

    def to(self, device):
        super().to(device)
        for pipeline in (self.renderer, self.viz_renderer):  # cameras and lights are not modules: moved by hand
            pipeline.to(device)
        return self

    def update_lights(self, direction=None, ka=None, kd=None, ks=None):
        given = dict(direction=direction, ka=ka, kd=kd, ks=ks)
        for arg, attr in LIGHT_FIELDS:
            if given[arg] is not None:
                setattr(self.lights, attr, given[arg])

    def reset_default_lights(self):
        for attr, value in self._light_defaults.items():
            setattr(self.lights, attr, value)

    @torch.no_grad()
    def compute_vertex_visibility(self, meshes, R, T):
        """(N,V) bool: vertices of faces that own at least one pixel of the hard 2x-supersampled raster
        (renderer.py:105-116)."""
        rasterizer = self.viz_renderer.rasterizer
        rs = rasterizer.raster_settings
        face_verts, first, nf = rasterizer.transform(meshes, R, T)
        # only the face indices are wanted: depths, barycentrics and distances are not written at all
        p2f = native.rasterize_meshes(face_verts.detach(), first, nf, rs.image_size, rs.blur_radius, rs.faces_per_pixel,
                                      rs.clip_barycentric_coords, rs.cull_backfaces, want_zbuf=False, want_bary=False,
                                      want_dists=False)[0]
        packed_faces = meshes.faces_packed()
        n_verts = meshes.verts_packed().shape[0]
        face_visible = torch.zeros(packed_faces.shape[0] + 1, dtype=torch.bool, device=p2f.device)
        face_visible[(p2f.flatten() + 1).long()] = True  # slot 0 collects the -1 entries
        vis = torch.zeros(n_verts, dtype=torch.bool, device=p2f.device)
        vis[packed_faces[face_visible[1:]].flatten()] = True
        return vis.view(len(meshes), -1)


@torch.no_grad()
def render_rotated_views(mesh, img_size=256, n_views=50, elev=30, dist=2.5, R=None, T=None, bkg=None, renderer=None,
                         rend_kwargs=None, eye_light=False, device=None):
    """(n_views, 3, img_size, img_size) turntable of ONE mesh through the visualisation path, composited over `bkg` if
    given -- the reference's helper of the same name (renderer.py:219-257), which `save_mesh_as_gif` wraps.  With
    `eye_light` an ambient-lit renderer is rebuilt with a headlight (directional, along the view axis, phong, K=1)
    that follows the camera; a given (R, T) is applied on top of the orbit (then elev / dist are taken as 0)."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    renderer = renderer if renderer is not None else Renderer(img_size, **(rend_kwargs or {}))
    if eye_light:
        if R is not None:
            raise NotImplementedError
        if isinstance(renderer.lights, AmbientLights) and not hasattr(renderer.lights, "diffuse_color"):
            kw = renderer.init_kwargs
            kw.update(lights={"name": "directional", "direction": [[0, 0, -1]], "ambient_color": [[0.7, 0.7, 0.7]],
                              "diffuse_color": [[0.3, 0.3, 0.3]], "specular_color": [[0.0, 0.0, 0.0]]},
                      shading_type="phong", faces_per_pixel=1)
            renderer = Renderer(img_size, **kw)
    elev = 0 if R is not None else elev
    dist = 0 if T is not None else dist
    R = torch.eye(3, device=device) if R is None else R.to(device)
    T = torch.zeros(3, device=device) if T is None else T.to(device)
    if bkg is not None and bkg.shape[-1] != img_size:
        bkg = torch.nn.functional.interpolate(bkg[None], size=(img_size, img_size), mode="bilinear",
                                              align_corners=False)[0]
    mesh, renderer = mesh.to(device), renderer.to(device)
    azim = torch.linspace(-180, 180, n_views)
    views, per_call = [], 10  # (the reference's batch of views per call)
    for k in range(0, n_views, per_call):
        R_view = look_at_view_transform(dist=1, elev=elev, azim=azim[k:k + per_call], device=device)[0]
        n = R_view.shape[0]
        T_view = torch.tensor([[0.0, 0.0, float(dist)]], device=device).expand(n, -1)
        if eye_light:  # the light points where the camera looks: (n, 1, 3)
            renderer.update_lights(direction=torch.tensor([[0.0, 0.0, -1.0]], device=device) @ R_view.transpose(1, 2))
        views.append(renderer(mesh.extend(n), R_view @ R, T_view + T, viz_purpose=True).clamp(0, 1).cpu())
    rec, alpha = torch.cat(views, 0).split([3, 1], dim=1)
    return rec if bkg is None else rec * alpha + (1 - alpha) * bkg.cpu()

