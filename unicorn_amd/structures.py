"""Batched mesh / texture containers with the method surface the reference's callers use on pytorch3d's `Meshes`,
`TexturesUV` and `TexturesVertex` (unicorn.py:262-266,283,350,389,407,499,512; utils/mesh.py:21-139;
renderer.py:109-110,273).  Own implementation: tensors are kept padded (N,V,3)/(N,F,3) when every mesh of the batch
has the same size (the reference's only case: one icosphere topology), which is also the layout the HIP kernels read.
"""
import torch


def _as_list(x):
    if torch.is_tensor(x):
        if x.dim() != 3:
            raise ValueError("padded tensors must be (N, M, 3)")
        return list(x.unbind(0))
    return list(x)


class TexturesUV:
    """maps (N,Ht,Wt,C) in [0,1]; faces_uvs (N,F,3) int64 indices into verts_uvs (N,V,2).  Sampling follows
    pytorch3d 0.6.2: bilinear, align_corners=True, padding_mode='border', v axis flipped."""

    def __init__(self, maps, faces_uvs, verts_uvs, padding_mode="border", align_corners=True, sampling_mode="bilinear"):
        if padding_mode != "border" or not align_corners or sampling_mode != "bilinear":
            raise NotImplementedError("only bilinear / align_corners=True / border padding (pytorch3d defaults)")
        if not torch.is_tensor(maps):
            maps = torch.stack(list(maps))
        if not torch.is_tensor(faces_uvs):
            faces_uvs = torch.stack(list(faces_uvs))
        if not torch.is_tensor(verts_uvs):
            verts_uvs = torch.stack(list(verts_uvs))
        if maps.dim() != 4 or faces_uvs.dim() != 3 or verts_uvs.dim() != 3:
            raise ValueError("expected maps (N,H,W,C), faces_uvs (N,F,3), verts_uvs (N,V,2)")
        self._maps, self._faces_uvs, self._verts_uvs = maps, faces_uvs, verts_uvs
        self._num_faces_per_mesh = [faces_uvs.shape[1]] * len(maps)
        self.device = maps.device

    def __len__(self):
        return self._maps.shape[0]

    def maps_padded(self):
        return self._maps

    def faces_uvs_padded(self):
        return self._faces_uvs

    def verts_uvs_padded(self):
        return self._verts_uvs

    def maps_list(self):
        return list(self._maps.unbind(0))

    def _shared_tables(self):
        """True when faces_uvs / verts_uvs are batch-expanded views of one table (unicorn.py:283 builds them with
        .expand): one (F,3,2) face-UV table then serves every mesh."""
        return self._faces_uvs.stride(0) == 0 and self._verts_uvs.stride(0) == 0

    _uv_cache = {}  # (storage ptrs, versions, shapes) -> gathered face-UV table; see faces_verts_uvs

    def faces_verts_uvs(self):
        """(face_uvs, shared): (F,3,2) with shared=True, else packed (N*F,3,2).  The reference rebuilds the textures
        object from the same two constant buffers every step (unicorn.py:283); the gathered table is kept for as long
        as those buffers are unchanged (same storage, same version counter)."""
        if self._shared_tables() or len(self) == 1:
            vu, fu = self._verts_uvs, self._faces_uvs
            if vu.requires_grad:
                return vu[0][fu[0]].contiguous(), True
            key = (vu.data_ptr(), vu._version, vu.shape, fu.data_ptr(), fu._version, fu.shape, vu.device)
            hit = TexturesUV._uv_cache.get(key)
            if hit is None:
                if len(TexturesUV._uv_cache) > 16:
                    TexturesUV._uv_cache.clear()
                # the entry keeps vu / fu alive, so their addresses cannot be handed to other tensors meanwhile
                hit = TexturesUV._uv_cache[key] = (vu[0][fu[0]].contiguous(), vu, fu)
            return hit[0], True
        idx = self._faces_uvs  # (N,F,3)
        N, F = idx.shape[:2]
        out = torch.gather(self._verts_uvs, 1, idx.reshape(N, F * 3, 1).expand(-1, -1, 2)).view(N * F, 3, 2)
        return out.contiguous(), False

    def detach(self):
        return TexturesUV(self._maps.detach(), self._faces_uvs.detach(), self._verts_uvs.detach())

    def clone(self):
        return TexturesUV(self._maps.clone(), self._faces_uvs.clone(), self._verts_uvs.clone())

    def to(self, device):
        return TexturesUV(self._maps.to(device), self._faces_uvs.to(device), self._verts_uvs.to(device))

    def extend(self, N):
        r = lambda t: t.repeat_interleave(N, dim=0)
        return TexturesUV(r(self._maps), r(self._faces_uvs), r(self._verts_uvs))

    def faces_verts_textures_packed(self):
        """(sum F, 3, C): the texture's colour at every face corner's uv (pytorch3d's method of the same name; the
        reference turns UV textures into vertex colours with it, utils/mesh.py:162-165).  Same sampling as
        sample_textures: bilinear, align_corners=True, border padding, v axis flipped."""
        N, F = self._faces_uvs.shape[:2]
        uv = torch.gather(self._verts_uvs, 1, self._faces_uvs.reshape(N, F * 3, 1).expand(-1, -1, 2))  # (N, F*3, 2)
        maps = torch.flip(self._maps.permute(0, 3, 1, 2), [2])
        tex = torch.nn.functional.grid_sample(maps, (uv * 2.0 - 1.0)[:, :, None, :], mode="bilinear", align_corners=True,
                                              padding_mode="border")                                  # (N, C, F*3, 1)
        return tex[..., 0].permute(0, 2, 1).reshape(N * F, 3, -1)

    def sample_textures(self, fragments, **kwargs):
        """texels (N,H,W,K,C) of the fragments (pytorch3d TexturesUV.sample_textures; what
        `meshes.sample_textures(fragments)` calls).  `fragments.pix_to_face` is packed over the batch this texture
        object belongs to."""
        from .shading import sample_textures

        class _Holder:  # the mesh side of the call is only the batch length for UV textures
            textures = self

            def __len__(_):
                return len(self)
        return sample_textures(_Holder(), fragments)

    def __getitem__(self, index):
        index = [index] if isinstance(index, int) else index
        return TexturesUV(self._maps[index], self._faces_uvs[index], self._verts_uvs[index])


class TexturesVertex:
    """Per-vertex features (N,V,C), interpolated with the barycentrics."""

    def __init__(self, verts_features):
        self._list = None  # only for a batch of meshes with different vertex counts (pytorch3d takes such a list)
        if not torch.is_tensor(verts_features):
            feats = list(verts_features)
            if len({f.shape[0] for f in feats}) > 1:
                self._list = feats
                V = max(f.shape[0] for f in feats)
                verts_features = torch.stack([torch.nn.functional.pad(f, (0, 0, 0, V - f.shape[0])) for f in feats])
            else:
                verts_features = torch.stack(feats)
        if verts_features.dim() != 3:
            raise ValueError("verts_features must be (N,V,C)")
        self._feat = verts_features
        self._num_verts_per_mesh = ([f.shape[0] for f in self._list] if self._list is not None
                                    else [verts_features.shape[1]] * len(verts_features))
        self.device = verts_features.device

    def __len__(self):
        return self._feat.shape[0]

    def verts_features_padded(self):
        return self._feat

    def verts_features_packed(self):
        if self._list is not None:
            return torch.cat(self._list, 0)
        return self._feat.reshape(-1, self._feat.shape[-1])

    def verts_features_list(self):
        return list(self._list) if self._list is not None else list(self._feat.unbind(0))

    def _map(self, fn):
        return TexturesVertex([fn(f) for f in self._list]) if self._list is not None else TexturesVertex(fn(self._feat))

    def detach(self):
        return self._map(lambda t: t.detach())

    def clone(self):
        return self._map(lambda t: t.clone())

    def to(self, device):
        return self._map(lambda t: t.to(device))

    def extend(self, N):
        if self._list is not None:
            return TexturesVertex([f for f in self._list for _ in range(N)])
        return TexturesVertex(self._feat.repeat_interleave(N, dim=0))

    def sample_textures(self, fragments, faces_packed=None):
        """vertex features interpolated with the barycentrics (pytorch3d TexturesVertex.sample_textures): needs the
        batch's packed faces"""
        if faces_packed is None:
            raise ValueError("TexturesVertex.sample_textures needs faces_packed")
        from .shading import interpolate_face_attributes
        return interpolate_face_attributes(fragments.pix_to_face, fragments.bary_coords,
                                           self.verts_features_packed()[faces_packed])

    def __getitem__(self, index):
        index = [index] if isinstance(index, int) else index
        if self._list is not None:
            if isinstance(index, slice):
                picked = self._list[index]
            else:
                index = torch.as_tensor(index)
                index = index.nonzero()[:, 0] if index.dtype == torch.bool else index
                picked = [self._list[int(i)] for i in index.reshape(-1)]
            return TexturesVertex(picked)
        return TexturesVertex(self._feat[index])


class Meshes:
    """Batch of triangle meshes.  verts: list of (V_n,3) float tensors or a padded (N,V,3) tensor; faces likewise
    (int64).  Autograd flows through the vertex tensors."""

    def __init__(self, verts, faces, textures=None):
        self._verts_list, self._faces_list = None, None
        self._verts_padded, self._faces_padded = None, None
        if torch.is_tensor(verts) and torch.is_tensor(faces):
            if verts.dim() != 3 or faces.dim() != 3:
                raise ValueError("padded verts / faces must be (N,V,3) / (N,F,3)")
            self._verts_padded, self._faces_padded = verts, faces
            self._N = verts.shape[0]
            self._equisized = True  # padded input is taken as unpadded same-size meshes
        else:
            self._verts_list, self._faces_list = _as_list(verts), _as_list(faces)
            self._N = len(self._verts_list)
            if len(self._faces_list) != self._N:
                raise ValueError("verts and faces must have the same batch size")
            self._equisized = (len({v.shape[0] for v in self._verts_list}) <= 1 and
                               len({f.shape[0] for f in self._faces_list}) <= 1)
        self.textures = textures
        self._cache = {}

    # ---- basic ----------------------------------------------------------------------------------------------
    def __len__(self):
        return self._N

    @property
    def device(self):
        return (self._verts_padded if self._verts_padded is not None else self._verts_list[0]).device

    @property
    def equisized(self):
        return self._equisized

    def isempty(self):
        return self._N == 0 or self.verts_packed().shape[0] == 0

    def verts_list(self):
        if self._verts_list is None:
            self._verts_list = list(self._verts_padded.unbind(0))
        return self._verts_list

    def faces_list(self):
        if self._faces_list is None:
            self._faces_list = list(self._faces_padded.unbind(0))
        return self._faces_list

    def num_verts_per_mesh(self):
        return torch.tensor([v.shape[0] for v in self.verts_list()], dtype=torch.int64, device=self.device)

    def num_faces_per_mesh(self):
        return torch.tensor([f.shape[0] for f in self.faces_list()], dtype=torch.int64, device=self.device)

    def verts_padded(self):
        if self._verts_padded is None:
            if self._equisized:
                self._verts_padded = torch.stack(self._verts_list) if self._N else torch.zeros(0, 0, 3)
            else:
                V = max(v.shape[0] for v in self._verts_list)
                self._verts_padded = torch.stack([torch.nn.functional.pad(v, (0, 0, 0, V - v.shape[0]))
                                                  for v in self._verts_list])
        return self._verts_padded

    def faces_padded(self):
        if self._faces_padded is None:
            if self._equisized:
                self._faces_padded = torch.stack(self._faces_list) if self._N else torch.zeros(0, 0, 3).long()
            else:
                F = max(f.shape[0] for f in self._faces_list)
                self._faces_padded = torch.stack([torch.nn.functional.pad(f, (0, 0, 0, F - f.shape[0]), value=-1)
                                                  for f in self._faces_list])
        return self._faces_padded

    def verts_packed(self):
        if self._equisized:
            return self.verts_padded().reshape(-1, 3)
        return torch.cat(self.verts_list(), 0)

    def mesh_to_verts_packed_first_idx(self):
        nv = self.num_verts_per_mesh()
        return torch.cumsum(nv, 0) - nv

    def mesh_to_faces_packed_first_idx(self):
        nf = self.num_faces_per_mesh()
        return torch.cumsum(nf, 0) - nf

    def faces_packed(self):
        if "faces_packed" not in self._cache:
            first = self.mesh_to_verts_packed_first_idx()
            if self._equisized:
                fp = (self.faces_padded() + first.view(-1, 1, 1)).reshape(-1, 3)
            else:
                fp = torch.cat([f + o for f, o in zip(self.faces_list(), first)], 0)
            self._cache["faces_packed"] = fp
        return self._cache["faces_packed"]

    def verts_packed_to_mesh_idx(self):
        return torch.repeat_interleave(torch.arange(self._N, device=self.device), self.num_verts_per_mesh())

    def faces_packed_to_mesh_idx(self):
        return torch.repeat_interleave(torch.arange(self._N, device=self.device), self.num_faces_per_mesh())

    def faces_normals_packed(self):
        """unit face normals (NF,3): normalize(cross(v1 - v0, v2 - v0), eps=1e-6) (pytorch3d Meshes)"""
        fv = self.verts_packed()[self.faces_packed()]
        n = torch.cross(fv[:, 1] - fv[:, 0], fv[:, 2] - fv[:, 0], dim=1)
        return torch.nn.functional.normalize(n, eps=1e-6, dim=1)

    def verts_normals_packed(self):
        """area-weighted vertex normals (V,3): every face adds the cross product of its two edges at each corner,
        then normalize(eps=1e-6) (pytorch3d Meshes._compute_vertex_normals)"""
        verts, faces = self.verts_packed(), self.faces_packed()
        fv = verts[faces]
        out = torch.zeros_like(verts)
        out = out.index_add(0, faces[:, 1], torch.cross(fv[:, 2] - fv[:, 1], fv[:, 0] - fv[:, 1], dim=1))
        out = out.index_add(0, faces[:, 2], torch.cross(fv[:, 0] - fv[:, 2], fv[:, 1] - fv[:, 2], dim=1))
        out = out.index_add(0, faces[:, 0], torch.cross(fv[:, 1] - fv[:, 0], fv[:, 2] - fv[:, 0], dim=1))
        return torch.nn.functional.normalize(out, eps=1e-6, dim=1)

    def get_mesh_verts_faces(self, index):
        return self.verts_list()[index], self.faces_list()[index]

    def _edges(self):
        if "edges" not in self._cache:
            f = self.faces_packed()
            e = torch.cat([f[:, [1, 2]], f[:, [2, 0]], f[:, [0, 1]]], 0)  # edge i is opposite to corner i
            e, _ = e.sort(dim=1)
            V = max(int(self.verts_packed().shape[0]), 1)
            key = e[:, 0] * V + e[:, 1]
            uniq, inv = torch.unique(key, sorted=True, return_inverse=True)
            edges = torch.stack([uniq // V, uniq % V], 1)
            F = f.shape[0]
            self._cache["edges"] = (edges, inv.view(3, F).t().contiguous())
        return self._cache["edges"]

    def edges_packed(self):
        return self._edges()[0]

    def faces_packed_to_edges_packed(self):
        return self._edges()[1]

    # ---- transforms -----------------------------------------------------------------------------------------
    def _new_verts(self, verts_padded=None, verts_list=None, textures="same"):
        tex = self.textures if textures == "same" else textures
        if verts_padded is not None:
            faces = self._faces_padded if self._faces_padded is not None else self.faces_padded()
            m = Meshes(verts_padded, faces, tex)
        else:
            m = Meshes(verts_list, self.faces_list(), tex)
        for k in ("faces_packed", "edges"):
            if k in self._cache:
                m._cache[k] = self._cache[k]
        return m

    def _assign(self, other):
        self._verts_list, self._verts_padded = other._verts_list, other._verts_padded
        return self

    def offset_verts(self, offsets_packed):
        """offsets (sum V_n, 3) packed, or a (3,) vector"""
        if offsets_packed.shape == (3,):
            offsets_packed = offsets_packed.expand(self.verts_packed().shape[0], 3)
        if self._equisized:
            return self._new_verts(verts_padded=self.verts_padded() + offsets_packed.view(self.verts_padded().shape))
        nv = [v.shape[0] for v in self.verts_list()]
        return self._new_verts(verts_list=[v + o for v, o in zip(self.verts_list(), offsets_packed.split(nv))])

    def offset_verts_(self, offsets_packed):
        return self._assign(self.offset_verts(offsets_packed))

    def scale_verts(self, scale):
        """scale: float, (N,) or (N,3) tensor (unicorn.py:92,266 pass (1,3) / (B,3))"""
        if not torch.is_tensor(scale):
            scale = torch.full((self._N,), float(scale), device=self.device)
        if scale.shape[0] != self._N:
            scale = scale.expand(self._N, *scale.shape[1:])
        if self._equisized:
            s = scale.view(self._N, 1, -1) if scale.dim() == 2 else scale.view(self._N, 1, 1)
            return self._new_verts(verts_padded=self.verts_padded() * s)
        return self._new_verts(verts_list=[scale[i] * v for i, v in enumerate(self.verts_list())])

    def scale_verts_(self, scale):
        return self._assign(self.scale_verts(scale))

    def update_padded(self, new_verts_padded):
        return self._new_verts(verts_padded=new_verts_padded)

    def detach(self):
        tex = self.textures.detach() if self.textures is not None else None
        if self._equisized:
            return self._new_verts(verts_padded=self.verts_padded().detach(), textures=tex)
        return self._new_verts(verts_list=[v.detach() for v in self.verts_list()], textures=tex)

    def clone(self):
        tex = self.textures.clone() if self.textures is not None else None
        return Meshes([v.clone() for v in self.verts_list()], [f.clone() for f in self.faces_list()], tex)

    def to(self, device):
        tex = self.textures.to(device) if self.textures is not None else None
        if self._equisized and self._N:
            return Meshes(self.verts_padded().to(device), self.faces_padded().to(device), tex)
        return Meshes([v.to(device) for v in self.verts_list()], [f.to(device) for f in self.faces_list()], tex)

    def cuda(self):
        return self.to("cuda")

    def cpu(self):
        return self.to("cpu")

    def extend(self, N):
        """N copies of every mesh, repeat_interleave order (m0, m0, ..., m1, m1, ...)"""
        tex = self.textures.extend(N) if self.textures is not None else None
        if self._equisized and self._N:
            faces = self.faces_padded()
            # a single mesh (unicorn.py:263 extends its one template): the copies share its face table -- an expanded
            # view, never written to -- so that the renderer's and the regularisers' shared-topology paths see ONE table
            # (the vertices are copied: offset_verts_ / scale_verts_ replace them anyway)
            faces = faces.expand(N, -1, -1) if self._N == 1 else faces.repeat_interleave(N, 0)
            return Meshes(self.verts_padded().repeat_interleave(N, 0), faces, tex)
        verts, faces = [], []
        for v, f in zip(self.verts_list(), self.faces_list()):
            verts += [v.clone() for _ in range(N)]
            faces += [f.clone() for _ in range(N)]
        return Meshes(verts, faces, tex)

    def __getitem__(self, index):
        if isinstance(index, int):
            index = [index]
        tex = self.textures[index] if self.textures is not None else None
        if isinstance(index, slice):
            return Meshes(self.verts_list()[index], self.faces_list()[index], tex)
        if torch.is_tensor(index):
            index = index.nonzero().flatten().tolist() if index.dtype == torch.bool else index.tolist()
        return Meshes([self.verts_list()[i] for i in index], [self.faces_list()[i] for i in index], tex)

    def sample_textures(self, fragments):
        from .shading import sample_textures
        return sample_textures(self, fragments)


def _same_expanded_table(tensors):
    """the one table all of `tensors` are batch-expanded views of (stride 0 over the batch, same memory), or None"""
    t0 = tensors[0]
    if all(t.dim() == 3 and t.stride(0) == 0 and t.data_ptr() == t0.data_ptr() and t.shape[1:] == t0.shape[1:] and
           t.stride()[1:] == t0.stride()[1:] for t in tensors):
        return t0[:1]
    return None


def _cat_batches(tensors):
    """torch.cat along the batch -- or, when every input is an expanded view of ONE table (the reference builds faces /
    faces_uvs / verts_uvs with .expand, unicorn.py:262-266,283), that table expanded over the joined batch: the
    renderer's shared-table fast paths survive the join"""
    table = _same_expanded_table(tensors)
    if table is not None:
        return table.expand(sum(t.shape[0] for t in tensors), -1, -1)
    return torch.cat(tensors, 0)


def join_meshes_as_batch(meshes, include_textures=True):
    """pytorch3d.structures.join_meshes_as_batch: one batch holding the meshes of all the given batches, in order,
    textures joined too (all TexturesUV with maps of one size, or all TexturesVertex, or none).  Equal-sized inputs stay
    padded tensors (one cat each); gradients flow back to every input.  What the 3D step needs to render its three
    independent 32-mesh batches (unicorn.py:233, 411-413) in ONE renderer call."""
    meshes = list(meshes)
    if not meshes:
        raise ValueError("no meshes to join")
    if len(meshes) == 1:
        return meshes[0]
    tex = None
    if include_textures:
        texs = [m.textures for m in meshes]
        if all(isinstance(t, TexturesUV) for t in texs):
            if len({tuple(t.maps_padded().shape[1:]) for t in texs}) != 1:
                raise ValueError("texture maps of different sizes cannot be joined")
            tex = TexturesUV(torch.cat([t.maps_padded() for t in texs], 0),
                             _cat_batches([t.faces_uvs_padded() for t in texs]),
                             _cat_batches([t.verts_uvs_padded() for t in texs]))
        elif all(isinstance(t, TexturesVertex) for t in texs):
            tex = TexturesVertex(torch.cat([t.verts_features_padded() for t in texs], 0))
        elif any(t is not None for t in texs):
            raise ValueError("all meshes must have textures of one kind (or none) to be joined with them")
    if all(m.equisized and len(m) > 0 for m in meshes):
        vs, fs = [m.verts_padded() for m in meshes], [m.faces_padded() for m in meshes]
        if len({v.shape[1] for v in vs}) == 1 and len({f.shape[1] for f in fs}) == 1:
            return Meshes(torch.cat(vs, 0), _cat_batches(fs), tex)
    verts, faces = [], []
    for m in meshes:
        verts += m.verts_list()
        faces += m.faces_list()
    return Meshes(verts, faces, tex)


class TiledTexturesUV(TexturesUV):
    """The textures of `repeat(mesh, reps)`: the base textures `reps` times in tile order, WITHOUT the copy until a
    caller asks for the repeated tensors (the fused renderer never does: it indexes the base maps)."""

    def __init__(self, base, reps):
        self.tile_base, self.tile_reps = base, reps
        self._tiled = None
        self._num_faces_per_mesh = base._num_faces_per_mesh * reps
        self.device = base.device

    def _materialise(self):
        if self._tiled is None:
            b, n = self.tile_base, self.tile_reps
            rep = (lambda t: t[:1].expand(len(b) * n, -1, -1)) if b._shared_tables() else (lambda t: t.repeat(n, 1, 1))
            self._tiled = (b.maps_padded().repeat(n, 1, 1, 1), rep(b.faces_uvs_padded()), rep(b.verts_uvs_padded()))
        return self._tiled

    _maps = property(lambda self: self._materialise()[0])
    _faces_uvs = property(lambda self: self._materialise()[1])
    _verts_uvs = property(lambda self: self._materialise()[2])

    def __len__(self):
        return len(self.tile_base) * self.tile_reps


class TiledMeshes(Meshes):
    """`repeat(mesh, reps)` for equal-sized meshes: reps copies in torch `repeat` (tile) order -- mesh i of the result
    is base mesh i % B -- held as the base meshes plus the count.  The fused renderer reads the base tensors through
    that index (ur_render_meshes_* with n_meshes = B: no copy of verts, faces or texture maps, and the gradients of
    all poses of a mesh are summed by the kernels); any other consumer gets ordinary repeated tensors, made on first
    use."""

    def __init__(self, base, reps):
        self.tile_base, self.tile_reps = base, int(reps)
        self._N = len(base) * self.tile_reps
        self._equisized = True
        self._verts_list = self._faces_list = None
        self._tiled_verts = self._tiled_faces = None
        tex = base.textures
        if isinstance(tex, TexturesUV):
            self.textures = TiledTexturesUV(tex, self.tile_reps)
        elif isinstance(tex, TexturesVertex):
            self.textures = TexturesVertex(tex.verts_features_padded().repeat(self.tile_reps, 1, 1))
        elif tex is None:
            self.textures = None
        else:
            raise NotImplementedError
        self._cache = {}

    @property
    def _verts_padded(self):
        if self._tiled_verts is None:
            self._tiled_verts = self.tile_base.verts_padded().repeat(self.tile_reps, 1, 1)
        return self._tiled_verts

    @_verts_padded.setter
    def _verts_padded(self, value):  # offset_verts_ / scale_verts_ assign new vertices: the tiling ends there
        if self.tile_base is not None:
            _ = self._faces_padded  # (made from the base while it is still known)
        self._tiled_verts = value
        self.tile_base = None

    @property
    def _faces_padded(self):
        if self._tiled_faces is None:
            f = self.tile_base.faces_padded()
            # one shared topology stays a batch-expanded view (stride 0); per-mesh tables are repeated
            self._tiled_faces = f[:1].expand(self._N, -1, -1) if f.stride(0) == 0 else f.repeat(self.tile_reps, 1, 1)
        return self._tiled_faces

    @_faces_padded.setter
    def _faces_padded(self, value):
        self._tiled_faces = value

    @property
    def device(self):
        return self.tile_base.device if self.tile_base is not None else self._tiled_verts.device


def repeat(mesh, N):
    """N copies in torch `repeat` (tile) order -- the convention of the reference's utils/mesh.py:43-71 (candidate
    poses are K-major, unicorn.py:295-304).  Equal-sized meshes are not copied at all (TiledMeshes: the renderer
    indexes the B meshes with image % B); others take one list operation."""
    assert N >= 1
    if N == 1:
        return mesh
    if mesh.equisized and len(mesh) > 0:
        return TiledMeshes(mesh, N)
    tex = mesh.textures
    if isinstance(tex, TexturesVertex):
        new_tex = TexturesVertex(tex.verts_features_padded().repeat(N, 1, 1))
    elif isinstance(tex, TexturesUV):
        new_tex = TexturesUV(tex.maps_padded().repeat(N, 1, 1, 1), tex.faces_uvs_padded().repeat(N, 1, 1),
                             tex.verts_uvs_padded().repeat(N, 1, 1))
    elif tex is None:
        new_tex = None
    else:
        raise NotImplementedError
    return Meshes(mesh.verts_list() * N, mesh.faces_list() * N, new_tex)
