"""Generates tests/golden/*.npz.  Runs ONLY in the build container (needs /root/reference); the GPU box
uses the committed .npz files.

What is pinned here, and by what:
  G1 blend_*.npz      the reference's own `layered_rgb_blend` (src/model/renderer.py:175-202): the function's AST
                      node is taken from the reference file where it lies and executed on CPU torch tensors,
                      forward and autograd backward, for its three alpha modes (sigma == 0, clip_inside, sigmoid).
                      (The module itself cannot be imported: its top-level imports need pytorch3d)
  G2 uvsample.npz     torch.nn.functional.grid_sample(flip(map), 2*uv-1, bilinear, align_corners=True,
                      padding_mode='border') -- the ATen op pytorch3d's TexturesUV.sample_textures calls
                      (reached from renderer.py:162) -- forward and autograd backward.
  G3 helpers.npz      convert_3d_to_uv_coordinates, azim/elev/roll_to_rotation_matrix (src/model/tools.py:164-171,
                      197-231), AST-extracted the same way; pins unicorn_amd/synth.py.
  G4 regularizers.npz the reference's own `normal_consistency` (src/utils/mesh.py:92-139, icosphere / shared-topology
                      routine), AST-extracted and run on a duck-typed meshes object (verts / faces / edge tables are
                      fixture INPUTS, built by unicorn_amd.structures.Meshes on the CPU), value and autograd gradient;
                      plus the uniform Laplacian of pytorch3d's mesh_laplacian_smoothing RESTATED in float64 torch
                      (pytorch3d is absent: that half is not a reference output, parity unpinned).
  G5 lighting.npz     the reference's own `diffuse` and `specular` (src/model/pytorch3d_monkey.py:73-198, the functions
                      it patches into pytorch3d), AST-extracted.  Their only pytorch3d dependency is the helper
                      convert_to_tensors_and_broadcast, which turns its arguments into tensors of one batch size: every
                      argument is passed here as a tensor ALREADY broadcast to the points' shape (colour / direction /
                      camera position (N,H,W,K,3), shininess (N,)), on which that helper is the identity, so the
                      identity stands in for it and every later reshape branch of the functions is skipped.
Nothing of the reference's text is written to the fixtures: they hold inputs and outputs only.
"""
import ast
import os
import types

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference/src/model"
OUT = os.path.dirname(os.path.abspath(__file__))


def extract(path, names, extra_globals):
    tree = ast.parse(open(path).read())
    nodes = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(nodes) == len(names), [n.name for n in nodes]
    ns = dict(extra_globals)
    exec(compile(ast.Module(body=nodes, type_ignores=[]), path, "exec"), ns)
    return [ns[n] for n in names]


def make_blend():
    (blend,) = extract(os.path.join(REF, "renderer.py"), ["layered_rgb_blend"], {"torch": torch})
    N, H, W, K, NF = 2, 6, 5, 7, 11
    for seed, (name, sigma, clip_inside) in enumerate([("hard", 0.0, True), ("clip", 1e-4, True),
                                                       ("sigmoid", 1e-4, False), ("clip_s1e-2", 1e-2, True)]):
        g = torch.Generator().manual_seed(seed + 7)
        p2f = torch.randint(-1, NF, (N, H, W, K), generator=g)
        # padded entries come last in a real fragment list, but the blend must not rely on that
        scale = sigma if sigma > 0 else 1e-4
        dists = torch.randn(N, H, W, K, generator=g) * scale * 2
        dists[0, 0, 0, :] = 0.0  # exact zeros: boundary of clamp / (d <= 0)
        dists[0, 0, 1, :] = -dists[0, 0, 1, :].abs()  # everything inside
        dists[0, 0, 2, :] = dists[0, 0, 2, :].abs() * 30  # far outside -> alpha ~ 0
        colors = torch.rand(N, H, W, K, 3, generator=g)
        bgc = (1.0, 1.0, 1.0) if name != "sigmoid" else (0.2, 0.5, 0.9)
        cot = torch.randn(N, 4, H, W, generator=g)
        colors.requires_grad_(True)
        dists.requires_grad_(True)
        frag = types.SimpleNamespace(pix_to_face=p2f, dists=dists)
        bp = types.SimpleNamespace(sigma=sigma, background_color=bgc)
        img = blend(colors, frag, bp, clip_inside=clip_inside)
        (img * cot).sum().backward()
        gd = dists.grad if dists.grad is not None else torch.zeros_like(dists)
        # the function's debug outputs (renderer.py:199-200): alphas + trailing 1, transmittance in front of each layer
        with torch.no_grad():
            _, dbg_alpha, dbg_occ, dbg_img = blend(colors.detach(), frag, bp, clip_inside=clip_inside, debug=True)
        assert torch.equal(dbg_img, img.detach())
        np.savez(os.path.join(OUT, f"blend_{name}.npz"), pix_to_face=p2f.numpy(), dists=dists.detach().numpy(),
                 debug_alpha=dbg_alpha.numpy(), debug_occ_alpha=dbg_occ.numpy(),
                 colors=colors.detach().numpy(), background=np.array(bgc, np.float32), sigma=np.float64(sigma),
                 clip_inside=np.bool_(clip_inside), image=img.detach().numpy(), cotangent=cot.numpy(),
                 grad_colors=colors.grad.numpy(), grad_dists=gd.numpy())


def make_uvsample():
    N, Ht, Wt, C, P = 2, 8, 6, 3, 40
    g = torch.Generator().manual_seed(11)
    maps = torch.rand(N, Ht, Wt, C, generator=g)
    uv = torch.rand(N, P, 1, 2, generator=g) * 1.2 - 0.1  # some outside [0,1] -> border padding
    uv[0, 0, 0] = torch.tensor([0.0, 0.0])
    uv[0, 1, 0] = torch.tensor([1.0, 1.0])
    uv[0, 2, 0] = torch.tensor([0.5, 0.5])
    uv[0, 3, 0] = torch.tensor([0.2, 1.0])  # lands exactly on texel columns/rows
    cot = torch.randn(N, C, P, 1, generator=g)
    maps.requires_grad_(True)
    uv.requires_grad_(True)
    tex = F.grid_sample(torch.flip(maps.permute(0, 3, 1, 2), [2]), uv * 2.0 - 1.0, mode="bilinear", align_corners=True,
                        padding_mode="border")
    (tex * cot).sum().backward()
    np.savez(os.path.join(OUT, "uvsample.npz"), maps=maps.detach().numpy(), uv=uv.detach().numpy()[:, :, 0],
             texels=tex.detach().numpy()[..., 0].transpose(0, 2, 1), cotangent=cot.numpy()[..., 0].transpose(0, 2, 1),
             grad_maps=maps.grad.numpy(), grad_uv=uv.grad.numpy()[:, :, 0])


def make_helpers():
    fns = extract(os.path.join(REF, "tools.py"),
                  ["convert_3d_to_uv_coordinates", "azim_to_rotation_matrix", "elev_to_rotation_matrix",
                   "roll_to_rotation_matrix"], {"torch": torch, "np": np})
    uv_fn, az_fn, el_fn, ro_fn = fns
    g = torch.Generator().manual_seed(5)
    X = torch.randn(50, 3, generator=g)
    X[0] = torch.tensor([0.0, 1.0, 0.0])
    X[1] = torch.tensor([0.0, -1.0, 0.0])
    ang = torch.rand(3, 9, generator=g) * 360 - 180
    R = az_fn(ang[0]) @ el_fn(ang[1]) @ ro_fn(ang[2])  # unicorn.py:306
    np.savez(os.path.join(OUT, "helpers.npz"), X=X.numpy(), uv=uv_fn(X).numpy(), angles=ang.numpy(),
             R_azim=az_fn(ang[0]).numpy(), R_elev=el_fn(ang[1]).numpy(), R_roll=ro_fn(ang[2]).numpy(), R=R.numpy())


def make_regularizers():
    import sys
    sys.path.insert(0, os.path.join(OUT, "..", ".."))
    from unicorn_amd import synth
    from unicorn_amd.structures import Meshes
    (nc,) = extract("/root/reference/src/utils/mesh.py", ["normal_consistency"],
                    {"torch": torch, "mesh_normal_consistency": None})
    sc = synth.make_scene(3, level=2, tex_size=8, seed=11)
    verts = (sc["verts"].double() + 0.03 * torch.randn(sc["verts"].shape, generator=torch.Generator().manual_seed(3),
                                                       dtype=torch.float64)).requires_grad_(True)
    N, V = verts.shape[:2]
    faces = sc["faces"]
    m = Meshes(verts=verts, faces=faces[None].expand(N, -1, -1))
    duck = types.SimpleNamespace(isempty=lambda: False, device=verts.device, verts_packed=m.verts_packed,
                                 faces_packed=m.faces_packed, edges_packed=m.edges_packed,
                                 verts_packed_to_mesh_idx=m.verts_packed_to_mesh_idx,
                                 faces_packed_to_edges_packed=m.faces_packed_to_edges_packed)
    duck_cls = type("Duck", (), {"__len__": lambda self: N, "__getattr__": lambda self, k: getattr(duck, k)})
    loss = nc(duck_cls())
    (g_nc,) = torch.autograd.grad(loss, verts)
    # uniform Laplacian, restated: L v_i = mean_{j in N(i)} v_j - v_i ; loss = mean over meshes of mean_i ||L v_i||
    e = m.edges_packed()
    Vt = N * V
    A = torch.zeros(Vt, Vt, dtype=torch.float64)
    A[e[:, 0], e[:, 1]] = 1.0
    A[e[:, 1], e[:, 0]] = 1.0
    deg = A.sum(1, keepdim=True)
    Lv = (A / deg) @ m.verts_packed() - m.verts_packed()
    lap = (Lv.norm(dim=1) / V).sum() / N
    (g_lap,) = torch.autograd.grad(lap, verts)
    np.savez(os.path.join(OUT, "regularizers.npz"), verts=verts.detach().numpy(), faces=faces.numpy(),
             edges_packed=e.numpy(), faces_to_edges=m.faces_packed_to_edges_packed().numpy(),
             normal=loss.item(), grad_normal=g_nc.numpy(), laplacian=lap.item(), grad_laplacian=g_lap.numpy())


def make_lighting():
    ident = lambda *args, device=None: args  # convert_to_tensors_and_broadcast on already-broadcast tensors
    dif_fn, spe_fn = extract(os.path.join(REF, "pytorch3d_monkey.py"), ["diffuse", "specular"],
                             {"torch": torch, "F": F, "convert_to_tensors_and_broadcast": ident})
    N, H, W, K = 2, 4, 3, 2
    g = torch.Generator().manual_seed(23)
    shape = (N, H, W, K, 3)
    points = torch.randn(shape, generator=g, dtype=torch.float64)
    normals = torch.randn(shape, generator=g, dtype=torch.float64)
    normals[0, 0, 0, 0] = 0.0                                    # a padded fragment: zero normal (eps of normalize)
    normals[0, 0, 1, 0] = normals[0, 0, 1, 0] * 1e-8             # tiny: below the eps
    direction = torch.randn(N, 1, 1, 1, 3, generator=g, dtype=torch.float64).expand(shape).contiguous()
    direction[1] = (torch.randn(1, 3, generator=g, dtype=torch.float64) - points[1])  # a point light: per-point direction
    color = torch.rand(N, 1, 1, 1, 3, generator=g, dtype=torch.float64).expand(shape).contiguous()
    cam = (torch.randn(N, 1, 1, 1, 3, generator=g, dtype=torch.float64) * 3).expand(shape).contiguous()
    shin = torch.tensor([64.0, 8.0], dtype=torch.float64)       # (N,): the functions view it as (N,1,1,1) themselves
    for t in (points, normals, direction):
        t.requires_grad_(True)
    d = dif_fn(normals=normals, color=color, direction=direction)
    sp = spe_fn(points=points, normals=normals, direction=direction, color=color, camera_position=cam, shininess=shin)
    cot_d = torch.randn(shape, generator=g, dtype=torch.float64)
    cot_s = torch.randn(shape, generator=g, dtype=torch.float64)
    gn_d, gdir_d = torch.autograd.grad((d * cot_d).sum(), (normals, direction), retain_graph=True)
    gp_s, gn_s, gdir_s = torch.autograd.grad((sp * cot_s).sum(), (points, normals, direction))
    np.savez(os.path.join(OUT, "lighting.npz"), points=points.detach().numpy(), normals=normals.detach().numpy(),
             direction=direction.detach().numpy(), color=color.numpy(), camera_position=cam.numpy(),
             shininess=shin.numpy(), diffuse=d.detach().numpy(), specular=sp.detach().numpy(),
             cot_diffuse=cot_d.numpy(), cot_specular=cot_s.numpy(), diffuse_grad_normals=gn_d.numpy(),
             diffuse_grad_direction=gdir_d.numpy(), specular_grad_points=gp_s.numpy(),
             specular_grad_normals=gn_s.numpy(), specular_grad_direction=gdir_s.numpy())


if __name__ == "__main__":
    make_lighting()
    make_regularizers()
    make_blend()
    make_uvsample()
    make_helpers()
    print("wrote", sorted(f for f in os.listdir(OUT) if f.endswith(".npz")))
