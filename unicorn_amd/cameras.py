"""Cameras with the surface the reference uses on pytorch3d's FoVPerspectiveCameras / PerspectiveCameras
(renderer.py:59-63; kp_eval.py:91-104 `transform_points_screen`; renderer.py:247 `look_at_view_transform`).

Projection model: view = X @ R + T (row vectors); x_ndc = s*x/z, y_ndc = s*y/z, depth kept as
view z; s = 1/tan(fov/2) for FoV cameras (aspect ratio 1) or focal_length for Perspective cameras (principal point
0, NDC units).  +X points left and +Y up in NDC, as in pytorch3d.
"""
import math

import torch


class _Cameras:
    def __init__(self, device="cpu"):
        self.device = torch.device(device)

    def to(self, device):
        self.device = torch.device(device)
        return self

    def focal_scale(self):
        raise NotImplementedError

    def transform_points_view(self, points, R, T):
        return points @ R + T[:, None]

    def transform_points_ndc(self, points, R, T):
        """points (N,P,3) world -> (N,P,3) NDC xy + view depth"""
        sx, sy = self.focal_scale()
        v = self.transform_points_view(points, R, T)
        z = v[..., 2]
        return torch.stack([sx * v[..., 0] / z, sy * v[..., 1] / z, z], -1)

    def transform_points_screen(self, points, image_size=None, R=None, T=None, **kwargs):
        """NDC -> pixel coordinates, the inverse of the rasterizer's pixel-centre map (column xo has
        x_ndc = 1 - (2*xo+1)/W).  image_size: (H, W) or ((H, W),)."""
        if points.dim() == 2:
            points = points[None]
        N = points.shape[0]
        R = torch.eye(3, device=points.device).expand(N, 3, 3) if R is None else R
        T = torch.zeros(N, 3, device=points.device) if T is None else T
        ndc = self.transform_points_ndc(points, R, T)
        size = torch.as_tensor(image_size).flatten().tolist()
        H, W = (size[0], size[1]) if len(size) >= 2 else (size[0], size[0])
        s = min(H, W) / 2.0
        x = (W - 1) / 2.0 - ndc[..., 0] * s
        y = (H - 1) / 2.0 - ndc[..., 1] * s
        return torch.stack([x, y, ndc[..., 2]], -1)


class FoVPerspectiveCameras(_Cameras):
    def __init__(self, znear=1.0, zfar=100.0, aspect_ratio=1.0, fov=60.0, degrees=True, device="cpu", **unused):
        super().__init__(device)
        self.znear, self.zfar, self.aspect_ratio, self.fov, self.degrees = znear, zfar, aspect_ratio, fov, degrees

    def focal_scale(self):
        fov = math.radians(self.fov) if self.degrees else self.fov
        s = 1.0 / math.tan(fov / 2.0)
        return s / self.aspect_ratio, s


class PerspectiveCameras(_Cameras):
    def __init__(self, focal_length=1.0, principal_point=((0.0, 0.0),), device="cpu", **unused):
        super().__init__(device)
        pp = torch.as_tensor(principal_point, dtype=torch.float32).flatten().tolist()
        if any(abs(p) > 0 for p in pp):
            raise NotImplementedError("principal_point != 0")
        fl = torch.as_tensor(focal_length, dtype=torch.float32).flatten().tolist()
        self.focal_length = (fl[0], fl[0]) if len(fl) == 1 else (fl[0], fl[1])
        self.znear, self.zfar = 1.0, 100.0

    def focal_scale(self):
        return self.focal_length


def look_at_view_transform(dist=1.0, elev=0.0, azim=0.0, degrees=True, device="cpu"):
    """Camera on a sphere of radius `dist` looking at the origin, +Y up -> (R (N,3,3), T (N,3)) in the
    X_view = X_world @ R + T convention."""
    dist, elev, azim = [torch.as_tensor(t, dtype=torch.float32, device=device).flatten() for t in (dist, elev, azim)]
    n = max(dist.numel(), elev.numel(), azim.numel())
    dist, elev, azim = [t.expand(n) for t in (dist, elev, azim)]
    if degrees:
        elev, azim = elev * (math.pi / 180.0), azim * (math.pi / 180.0)
    C = torch.stack([dist * torch.cos(elev) * torch.sin(azim), dist * torch.sin(elev),
                     dist * torch.cos(elev) * torch.cos(azim)], -1)
    up = torch.tensor([0.0, 1.0, 0.0], device=device).expand(n, 3)
    z = torch.nn.functional.normalize(-C, dim=-1, eps=1e-5)
    x = torch.nn.functional.normalize(torch.cross(up, z, dim=-1), dim=-1, eps=1e-5)
    y = torch.nn.functional.normalize(torch.cross(z, x, dim=-1), dim=-1, eps=1e-5)
    degenerate = (x.abs().sum(-1, keepdim=True) < 5e-3)  # looking straight along the up vector
    x = torch.where(degenerate, torch.nn.functional.normalize(torch.cross(y, z, dim=-1), dim=-1, eps=1e-5), x)
    R = torch.stack([x, y, z], dim=-1)  # columns = camera axes in world coordinates
    T = -(C[:, None] @ R)[:, 0]
    return R, T
