/*
 * unicorn_render.h -- C ABI of the MI355X (gfx950) differentiable mesh renderer.
 *
 * This is the drop-in boundary for the hot path behind monniert/unicorn's `Renderer`
 * (src/model/renderer.py:25-116).  In the reference stack that path crosses into native code through the pybind11
 * module `pytorch3d._C` (pytorch3d==0.6.2, environment.yml:21, not vendored in the reference) plus ATen's
 * grid_sampler, and the blend runs as ~15 torch kernels (renderer.py:175-202).  Every entry point below names the
 * reference interface it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory unless it says "host";
 *   - the caller allocates everything (outputs and workspace); nothing here calls hipMalloc/hipFree or syncs;
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream) of the CURRENT device;
 *   - return value 0 = success, negative = error (ur_last_error() gives the message, thread-local);
 *   - floating point is fp32, face indices are int32 (pytorch3d uses int64; the Python layer widens on demand);
 *   - pix_to_face holds PACKED face indices (mesh n's faces start at mesh_first_face[n]), -1 = empty slot;
 *   - the library is stateless and thread-safe (the opt-in ur_profile_* timing hook aside).
 */
#ifndef UNICORN_RENDER_H
#define UNICORN_RENDER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UR_OK 0
#define UR_ERR_INVALID_ARG (-1)
#define UR_ERR_UNSUPPORTED (-2)
#define UR_ERR_LAUNCH (-3)
#define UR_ERR_WORKSPACE (-4)

/* blend modes of layered_rgb_blend (renderer.py:186-191) */
#define UR_BLEND_HARD 0        /* sigma == 0        : alpha = (dist <= 0)                      */
#define UR_BLEND_CLIP_INSIDE 1 /* clip_inside=True  : alpha = exp(-max(dist, 0) / sigma)       */
#define UR_BLEND_SIGMOID 2     /* clip_inside=False : alpha = sigmoid(-dist / sigma)           */
#define UR_BLEND_SOFTMAX 3     /* ur_blend_colors only: pytorch3d softmax_rgb_blend            */

const char* ur_last_error(void);
int ur_abi_version(void);

/*
 * Opt-in kernel timing (no counterpart in the reference; for bench.py's roofline leg and for tuning).  Between
 * ur_profile_begin and ur_profile_end every launch of the two dominant kernels -- the fused forward rasterizer
 * (UR_PROFILE_RASTER_FWD) and the fused backward (UR_PROFILE_RENDER_BWD) -- is bracketed by a pair of HIP events
 * recorded on the stream the kernel is launched on, up to max_launches launches; ur_profile_end waits for them and
 * returns the durations in milliseconds (host arrays) in launch order.  Process-wide state, not for use under graph
 * capture; the rest of the library stays stateless.
 */
#define UR_PROFILE_OTHER 0
#define UR_PROFILE_RASTER_FWD 1
#define UR_PROFILE_RENDER_BWD 2
int ur_profile_begin(int max_launches);
int ur_profile_end(float* ms, int* kinds, int capacity, int* count);

/* Bytes of device workspace the raster / render entry points need: per-face records for `total_faces` packed faces,
 * the workgroup schedule (two ints per 16x16 bin) plus, when faces_per_pixel K exceeds what the LDS holds per pixel, a
 * spill area for the per-pixel top-K lists. */
size_t ur_workspace_bytes(int64_t total_faces, int N, int H, int W, int K);

/*
 * World -> view -> NDC of every face corner, fused with the face gather.
 * Replaces: MeshRasterizer.transform + cameras.transform_points + verts_packed[faces_packed]
 *           (pytorch3d, reached from renderer.py:48 with the cameras of renderer.py:59-63).
 *   view = v @ R[n] + T[n] ; x_ndc = sx * x / z ; y_ndc = sy * y / z ; z_ndc = z_view
 *   verts (N,V,3) ; faces int64, mesh n reads faces + n*faces_mesh_stride (elements): stride 0 = one (F,3) topology
 *   shared by the batch (unicorn.py:283 expands it), stride F*3 = per-mesh (N,F,3) ; R (N,3,3), T (N,3)
 *   -> face_verts (N*F,3,3)
 */
int ur_project_faces(const float* verts, const int64_t* faces, int64_t faces_mesh_stride, const float* R,
                     const float* T, float sx, float sy, int N, int V, int F, float* face_verts, void* stream);

/* Backward of ur_project_faces: grad_face_verts (N*F,3,3) -> grad_verts (N,V,3), grad_R (N,3,3), grad_T (N,3).
 * The outputs are overwritten (no need to clear them); any may be NULL.  One workgroup per mesh sums per vertex in
 * LDS (f64, no global atomics on grad_verts); meshes whose V*24 B exceed the LDS are split into vertex windows, one
 * workgroup each, whose R / T sums meet by a few float atomics. */
int ur_project_faces_backward(const float* verts, const int64_t* faces, int64_t faces_mesh_stride, const float* R,
                              const float* T, float sx, float sy, int N, int V, int F, const float* grad_face_verts,
                              float* grad_verts, float* grad_R, float* grad_T, void* stream);

/*
 * Per-pixel top-K rasterization.
 * Replaces: pytorch3d._C.rasterize_meshes(face_verts, mesh_to_face_first_idx, num_faces_per_mesh,
 *           clipped_faces_neighbor_idx, image_size, blur_radius, faces_per_pixel, bin_size, max_faces_per_bin,
 *           perspective_correct, clip_barycentric_coords, cull_backfaces), call site renderer.py:46-49,54-57.
 * There is no bin_size / max_faces_per_bin: binning is internal (16x16 bins, lists in LDS, processed in rounds of
 * 2048 faces, bins scheduled heaviest first) and cannot overflow; any faces_per_pixel is accepted.  perspective_correct must be 0 (the reference never enables it, renderer.py:47,55).
 *   face_verts (NF,3,3) ; mesh_first_face, mesh_num_faces (N) int64 ;
 *   outputs (N,H,W,K): pix_to_face int32, zbuf, dists fp32 ; bary (N,H,W,K,3).  Any of zbuf/bary/dists may be NULL.
 */
int ur_rasterize_meshes(const float* face_verts, const int64_t* mesh_first_face, const int64_t* mesh_num_faces, int N,
                        int64_t total_faces, int H, int W, float blur_radius, int K, int perspective_correct,
                        int clip_barycentric_coords, int cull_backfaces, int32_t* pix_to_face, float* zbuf,
                        float* bary, float* dists, void* workspace, size_t workspace_bytes, void* stream);

/*
 * ur_rasterize_meshes without any allocation on anybody's side: no counterpart in the reference.  For callers that
 * cannot allocate device memory -- code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc -- and therefore cannot provide the workspace either.  Same arguments, kernels and results (bit for bit) as
 * ur_rasterize_meshes, minus workspace / workspace_bytes: the face records, the bin schedule and the K-list spill live
 * in a static device array of UR_STATIC_SCRATCH_BYTES that every GPU the library runs on holds until the process
 * ends.  A call that would need more (ur_workspace_bytes(total_faces, N, H, W, K) > UR_STATIC_SCRATCH_BYTES) returns
 * UR_ERR_WORKSPACE and launches nothing.  Calls no hipMalloc / hipFree.  All calls on one GPU share that GPU's array:
 * calls that overlap on one GPU (two streams, two threads) race; serialise them.
 */
#define UR_STATIC_SCRATCH_BYTES (64u << 20)
int ur_rasterize_meshes_static(const float* face_verts, const int64_t* mesh_first_face, const int64_t* mesh_num_faces,
                               int N, int64_t total_faces, int H, int W, float blur_radius, int K,
                               int perspective_correct, int clip_barycentric_coords, int cull_backfaces,
                               int32_t* pix_to_face, float* zbuf, float* bary, float* dists, void* stream);

/*
 * Replaces: pytorch3d._C.rasterize_meshes_backward(face_verts, pix_to_face, grad_zbuf, grad_bary, grad_dists,
 *           perspective_correct, clip_barycentric_coords) -> grad_face_verts (NF,3,3), reached from trainer.py:151.
 * grad_face_verts must be zero-filled by the caller.  Any of grad_zbuf / grad_bary / grad_dists may be NULL (= zeros).
 */
int ur_rasterize_meshes_backward(const float* face_verts, const int32_t* pix_to_face, const float* grad_zbuf,
                                 const float* grad_bary, const float* grad_dists, int N, int64_t total_faces, int H,
                                 int W, int K, int perspective_correct, int clip_barycentric_coords,
                                 float* grad_face_verts, void* stream);

/*
 * Bin assignment export (tests / debugging): which faces can touch which 8x8 pixel tile, ascending packed index.
 * counts (N,TH,TW) int32 exact even past `cap`; lists (N,TH,TW,cap) int32, -1 padded.  TH = ceil(H/8), TW = ceil(W/8).
 * Same scan code as ur_rasterize_meshes.  Replaces: pytorch3d._C._rasterize_meshes_coarse (bin_faces).
 */
int ur_tile_faces(const float* face_verts, const int64_t* mesh_first_face, const int64_t* mesh_num_faces, int N,
                  int64_t total_faces, int H, int W, float blur_radius, int cull_backfaces, int cap, int32_t* counts,
                  int32_t* lists, void* workspace, size_t workspace_bytes, void* stream);

/*
 * UV texture sampling fused with the layered blend: fragments -> RGBA image (N,4,H,W).
 * Replaces: LayeredShader.forward (renderer.py:159-172) = TexturesUV.sample_textures (interp_face_attrs +
 *           grid_sample(flip_y(map), 2uv-1, bilinear, align_corners=True, 'border')) + ambient multiply (:163-164)
 *           + layered_rgb_blend (:175-202).
 *   maps (N,Ht,Wt,3) ; face_uvs (NF or F,3,2) with face_uvs_shared = 1 when one (F,3,2) table serves every mesh
 *   (then mesh_first_face gives the offset to subtract) ; background host float[3] ; ambient host float[3] or NULL.
 */
int ur_shade_blend(const int32_t* pix_to_face, const float* bary, const float* dists, const float* maps,
                   const float* face_uvs, int face_uvs_shared, const int64_t* mesh_first_face, int N, int H, int W,
                   int K, int Ht, int Wt, float sigma, const float* background, const float* ambient, int mode,
                   float* image, void* stream);

/* Backward of ur_shade_blend: grad_image (N,4,H,W) -> grad_maps (N,Ht,Wt,3) [zero-filled by the caller, float
 * atomics], grad_bary (N,H,W,K,3), grad_dists (N,H,W,K) [fully written].  grad_maps may be NULL. */
int ur_shade_blend_backward(const float* grad_image, const int32_t* pix_to_face, const float* bary,
                            const float* dists, const float* maps, const float* face_uvs, int face_uvs_shared,
                            const int64_t* mesh_first_face, int N, int H, int W, int K, int Ht, int Wt, float sigma,
                            const float* background, const float* ambient, int mode, float* grad_maps,
                            float* grad_bary, float* grad_dists, void* stream);

/*
 * Blending of per-fragment colours computed by the caller (lit colours of phong / flat / gouraud shading).
 * Replaces: layered_rgb_blend(colors, fragments, blend_params, clip_inside) (renderer.py:172,175-202) for
 *           kind = UR_BLEND_HARD / CLIP_INSIDE / SIGMOID, and pytorch3d's softmax_rgb_blend(colors, fragments,
 *           blend_params, znear, zfar), reached through SoftPhongShaderPlus (renderer.py:205-208, the
 *           layered_shader=False branch of renderer.py:38-43), for kind = UR_BLEND_SOFTMAX:
 *   prob = sigmoid(-dists/sigma)*mask ; A = 1 - prod(1-prob) ; z_inv = (zfar-zbuf)/(zfar-znear)*mask ;
 *   w = prob*exp((z_inv - max(z_inv_max,1e-10))/gamma) ; delta = max(exp((1e-10 - z_inv_max)/gamma), 1e-10) ;
 *   rgb = (sum w c + delta bg) / (sum w + delta).
 *   colors (N,H,W,K,3) ; pix_to_face / dists / zbuf (N,H,W,K) (zbuf may be NULL unless softmax) -> image (N,4,H,W).
 * Backward: grad_image -> grad_colors, grad_dists (fully written) and grad_zbuf (may be NULL; zero unless softmax).
 */
int ur_blend_colors(const float* colors, const int32_t* pix_to_face, const float* dists, const float* zbuf, int N,
                    int H, int W, int K, int kind, float sigma, float gamma, const float* background, float znear,
                    float zfar, float* image, void* stream);
int ur_blend_colors_backward(const float* grad_image, const float* colors, const int32_t* pix_to_face,
                             const float* dists, const float* zbuf, int N, int H, int W, int K, int kind, float sigma,
                             float gamma, const float* background, float znear, float zfar, float* grad_colors,
                             float* grad_dists, float* grad_zbuf, void* stream);

/*
 * Mesh regularisers of the 3D step, value and gradient in one launch (one workgroup per mesh, shared topology).
 * Replaces: normal_consistency(meshes) (utils/mesh.py:92-139, the icosphere / shared-topology routine) and pytorch3d's
 *           mesh_laplacian_smoothing(meshes, method='uniform'), both called at unicorn.py:340-343, and their autograd.
 *   verts (N,V,3) ; edge_quads (E,4) int32, 16-byte aligned: (v0, v1, a, b) = an edge and the third vertex of each of
 *   its two faces (closed manifold) ; nbr_offsets (V+1) / nbr_index: vertex adjacency in CSR form ;
 *   normal_sum[n] = sum_e 1 - cos(n0, n1) ; laplacian_sum[n] = sum_i || mean_{j in N(i)} v_j - v_i || ;
 *   grad_normal / grad_laplacian (N,V,3): their gradients w.r.t. verts (either may be NULL) ; scratch (N,V,3) floats.
 * The reference's means are normal_sum.sum() / (N E) and (laplacian_sum / V).sum() / N.
 */
int ur_mesh_regularizers(const float* verts, int N, int V, const int32_t* edge_quads, int E, const int32_t* nbr_offsets,
                         const int32_t* nbr_index, float* normal_sum, float* laplacian_sum, float* grad_normal,
                         float* grad_laplacian, float* scratch, void* stream);

/* Self-test (no counterpart in the reference): the kernels divide by per-face quantities through a 5-instruction
 * exact-division sequence; this compares it with the hardware's IEEE division on n pseudo-random operand pairs and
 * adds the number of mismatching quotients to *mismatches (device uint64, zeroed by the caller).  Expected: 0. */
int ur_selftest_division(uint64_t n, uint32_t seed, uint64_t* mismatches, void* stream);

/*
 * The blend's per-layer intermediates, for LayeredShader(debug=True) (renderer.py:185-200 returns them next to the
 * image): alpha (N,H,W,K+1) = the layers' alphas (the mode's formula, 0 where pix_to_face < 0) followed by a 1 for the
 * background; occ_alpha (N,H,W,K+1) = the transmittance in front of every layer, 1, (1-a_0), (1-a_0)(1-a_1), ...
 * Inspection only (no backward).
 */
int ur_blend_intermediates(const int32_t* pix_to_face, const float* dists, int N, int H, int W, int K, float sigma,
                           int mode, float* alpha, float* occ_alpha, void* stream);

/*
 * Fused training path: rasterize + UV-sample + layered blend in ONE launch, and its backward in ONE launch.
 * Replaces: MeshRenderer.forward = MeshRasterizer (renderer.py:48) + LayeredShader.forward (renderer.py:159-172) and
 *           the autograd of both (trainer.py:151).  Numerically identical to ur_rasterize_meshes + ur_shade_blend:
 * the blend stops at the first layer that hides everything behind it (alpha == 1 with zero d(alpha)/d(dist)), which
 * changes neither the image nor any gradient.  No (N,H,W,K) fragment tensors are materialised: the forward leaves
 * only `layers`, ur_layers_elems() int32 words, opaque to the caller: per 16x16-pixel bin and per wavefront of the
 * bin (4 x 64 pixels, every second column / row -- or, in launches too small to fill the chip that way, N x bins <=
 * 1024, 8 x 32 pixels, every second column / fourth row: a pure function of the sizes, the same in both calls of a
 * step) the layers that matter, one 32-byte row each (packed face index,
 * pixel, float bits of the transmittance in front of the layer, its alpha and its colour), the layers of a pixel front
 * to back, pixel after pixel, followed by the per-pixel layer counts.  Only the rows of layers that exist are ever
 * touched.  The backward also needs ur_backward_scratch_bytes() of uninitialised device scratch.
 *   workspace: ur_workspace_bytes(total_faces, N, H, W, K) bytes; the backward must be given the SAME buffer, untouched.
 *   grad_face_verts (NF,3,3) and grad_maps (N,Ht,Wt,3) must be zero-filled by the caller; either may be NULL
 *   (then that gradient is skipped).  The z column of grad_face_verts stays 0 (the layered blend never reads zbuf).
 */
size_t ur_layers_elems(int N, int H, int W, int K);
size_t ur_backward_scratch_bytes(int N, int H, int W, int K);

int ur_render_layered_forward(const float* face_verts, const int64_t* mesh_first_face, const int64_t* mesh_num_faces,
                              int N, int64_t total_faces, int H, int W, float blur_radius, int K,
                              int clip_barycentric_coords, int cull_backfaces, const float* maps,
                              const float* face_uvs, int face_uvs_shared, int Ht, int Wt, float sigma,
                              const float* background, const float* ambient, int mode, float* image, int32_t* layers,
                              void* workspace, size_t workspace_bytes, void* stream);

int ur_render_layered_backward(const float* grad_image, const int64_t* mesh_first_face,
                               const int64_t* mesh_num_faces, int N, int64_t total_faces, int H, int W,
                               float blur_radius, int K, int clip_barycentric_coords, const float* maps,
                               const float* face_uvs, int face_uvs_shared, int Ht, int Wt, float sigma,
                               const float* background, const float* ambient, int mode, const int32_t* layers,
                               const void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes,
                               float* grad_face_verts, float* grad_maps, void* stream);

/*
 * The whole render step of a batch of EQUAL-SIZED meshes (the reference's case: one template topology, unicorn.py:86-93)
 * in one call each way: camera + face gather + rasterizer + shader.
 * Replaces: MeshRenderer.forward (renderer.py:48-49: MeshRasterizer.transform + rasterize_meshes + LayeredShader) as
 *           called by Renderer.forward (renderer.py:80-81), and its autograd (trainer.py:151).
 * Same numbers as ur_project_faces + ur_render_layered_forward (bit for bit) and their backwards.  Up to 32768 faces
 * per mesh everything before the schedule is ONE launch (a workgroup per 256 faces of a mesh: projection, face records,
 * per-bin load and 256-bit face masks in LDS -- no global atomics, nothing to pre-zero; the rasterizer then takes a
 * bin's face list from the masks instead of scanning the mesh's boxes); larger meshes go through the per-face kernels
 * inside the same call.
 *   verts (N,V,3), faces / faces_mesh_stride as in ur_project_faces, mesh_first_face[n] = n*F, mesh_num_faces[n] = F ;
 *   face_verts (N*F,3,3): OUTPUT (the projected corners; the caller may want them, the backward does not) ;
 *   zero_grad_face_verts (N*F,3,3) / zero_grad_maps (N,Ht,Wt,3): optional buffers the forward clears on the way, to be
 *   handed to ur_render_meshes_backward with grads_are_zero = 1 (saves the backward a fill launch); NULL = skip.  Of
 *   the first only the part the backward adds into is cleared: its first N*V*2 floats (per-vertex sums) when
 *   V*2 <= F*9, else all of it.
 * Backward: grad_image (N,4,H,W) -> grad_maps (N,Ht,Wt,3), grad_verts (N,V,3), grad_R (N,3,3), grad_T (N,3); any may
 *   be NULL.  grad_face_verts (N*F,3,3 floats) is the intermediate accumulator (needed when any of grad_verts /
 *   grad_R / grad_T is wanted); its contents after the call are unspecified (this call keeps per-vertex sums in it,
 *   not per-corner gradients: use ur_render_layered_backward for those).  grads_are_zero = 0 makes the call clear
 *   grad_face_verts and grad_maps itself.
 *   layers / workspace: what the forward left; scratch: ur_backward_scratch_bytes() of uninitialised memory.
 *
 * Copy-free pose tiling (n_meshes).  The reference renders every mesh under several candidate poses by physically
 * tiling the batch first: utils.mesh.repeat(meshes, K) clones verts / faces / texture maps K times (utils/mesh.py:43-71,
 * called at unicorn.py:232-233 before the renderer; poses are K-major, unicorn.py:295-304).  Here the N images may
 * instead SHARE n_meshes = M meshes: image n shows mesh n % M under pose (R[n], T[n]).  Then verts is (M,V,3), faces
 * has M (or one shared) tables, maps is (M,Ht,Wt,3), a per-mesh face_uvs table has M*F rows, and the backward adds
 * straight into grad_maps (M,Ht,Wt,3) and grad_verts (M,V,3) (the poses of a mesh summed in ascending order); R, T,
 * image, grad_R, grad_T, face_verts, layers and the accumulator stay per image.  n_meshes = 0 or N: one mesh per image.
 * N must be a multiple of M.  Tiling needs meshes the per-mesh kernels take -- ur_render_meshes_can_tile(V, F, H, W)
 * != 0: at most 32768 faces, at most 4096 bins, V*2 <= F*9 -- otherwise the calls return UR_ERR_UNSUPPORTED.
 */
int ur_render_meshes_can_tile(int V, int F, int H, int W);

int ur_render_meshes_forward(const float* verts, const int64_t* faces, int64_t faces_mesh_stride, const float* R,
                             const float* T, float sx, float sy, int N, int n_meshes, int V, int F,
                             const int64_t* mesh_first_face,
                             const int64_t* mesh_num_faces, int H, int W, float blur_radius, int K,
                             int clip_barycentric_coords, int cull_backfaces, const float* maps, const float* face_uvs,
                             int face_uvs_shared, int Ht, int Wt, float sigma, const float* background,
                             const float* ambient, int mode, float* face_verts, float* image, int32_t* layers,
                             float* zero_grad_face_verts, float* zero_grad_maps, void* workspace,
                             size_t workspace_bytes, void* stream);

int ur_render_meshes_backward(const float* grad_image, const float* verts, const int64_t* faces,
                              int64_t faces_mesh_stride, const float* R, const float* T, float sx, float sy, int N,
                              int n_meshes, int V, int F, const int64_t* mesh_first_face,
                              const int64_t* mesh_num_faces, int H, int W,
                              float blur_radius, int K, int clip_barycentric_coords, const float* maps,
                              const float* face_uvs, int face_uvs_shared, int Ht, int Wt, float sigma,
                              const float* background, const float* ambient, int mode, const int32_t* layers,
                              const void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes,
                              float* grad_face_verts, float* grad_maps, int grads_are_zero, float* grad_verts,
                              float* grad_R, float* grad_T, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UNICORN_RENDER_H */
