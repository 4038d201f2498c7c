// =====================================================================================
// oracle/unicorn_oracle.cpp -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.
//
// Single-threaded CPU restatement of the differentiable-rendering hot path behind
// monniert/unicorn's `Renderer` (reference: src/model/renderer.py:25-202).  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
// as the checker / the timed CPU baseline.  unicorn_amd/ never imports it.
//
// What each block follows:
//   * project_verts      cameras composed at renderer.py:59-63 (FoV / Perspective cameras of the
//                        un-vendored dependency pytorch3d==0.6.2, environment.yml:21) and the
//                        MeshRasterizer.transform call reached from renderer.py:48.
//   * rasterize_naive    the per-pixel top-K rasterizer reached from renderer.py:46-49,54-57
//                        (pytorch3d 0.6.2 `rasterize_meshes`, bin_size=0 algorithm class: for each
//                        pixel loop over all faces of the mesh, keep the K nearest by depth).
//   * rasterize_backward autograd of the above, reached from trainer.py:151.
//   * sample_textures    TexturesUV.sample_textures reached from renderer.py:162 with the textures
//                        built at unicorn.py:283 (maps NHWC, faces_uvs = faces, verts_uvs = uvs).
//   * layered_blend      layered_rgb_blend, renderer.py:175-202 (op order: clamp -> div -> exp ->
//                        mask -> cumprod -> weighted sum), all three alpha modes (:186-191).
//   * softmax_blend      pytorch3d softmax_rgb_blend reached through SoftPhongShaderPlus,
//                        renderer.py:205-208 (non-default `layered_shader=False` branch, :38-43).
//
// PARITY STATUS
//   * layered_blend and the bilinear UV sampling are PINNED: tests/golden/*.npz hold outputs of the
//     reference's own layered_rgb_blend (AST-extracted from renderer.py and executed in the build
//     container) and of torch.nn.functional.grid_sample (the exact ATen op TexturesUV calls);
//     tests/test_oracle_golden.py checks this file against them.
//   * The rasterizer is "PARITY UNPINNED": pytorch3d 0.6.2 is a third-party dependency absent from
//     /root/reference and from this image, and the reference holds no tests or golden vectors for
//     it.  The algorithm below restates its published behaviour (SURVEY.md section 8a spec block);
//     every place where a choice had to be made is marked SPEC DECISION.  It is anchored by analytic
//     known-answer tests and float64 finite-difference gradient checks (tests/test_oracle_raster.py).
//
// Build: g++ -O2 -ffp-contract=off -std=c++17 -shared -fPIC (see oracle/Makefile).  fp contraction
// is off so that fp32 results are reproducible op-by-op by the HIP kernels (bit-exact pix_to_face).
// =====================================================================================
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace {

template <typename T> struct V2 { T x, y; };

template <typename T> constexpr T kEps() { return T(1e-8); }

// edge(p, a, b) = (p.x-a.x)*(b.y-a.y) - (p.y-a.y)*(b.x-a.x)           SURVEY 8a spec block
template <typename T> inline T edge_fn(V2<T> p, V2<T> a, V2<T> b) {
  return (p.x - a.x) * (b.y - a.y) - (p.y - a.y) * (b.x - a.x);
}

// d edge / d(p, a, b), scaled by upstream g, accumulated.
template <typename T>
inline void edge_fn_bwd(V2<T> p, V2<T> a, V2<T> b, T g, V2<T>& gp, V2<T>& ga, V2<T>& gb) {
  gp.x += g * (b.y - a.y);
  gp.y += -g * (b.x - a.x);
  ga.x += g * (p.y - b.y);
  ga.y += g * (b.x - p.x);
  gb.x += -g * (p.y - a.y);
  gb.y += g * (p.x - a.x);
}

// pixel centre in NDC.  +X left / +Y up: output column xo maps to NDC index W-1-xo.
// SPEC DECISION (non-square): range = 2*S1/S2 when S1 > S2 else 2 (float division).
template <typename T> inline T pix_to_ndc(int i, int S1, int S2) {
  T range = T(2);
  if (S1 > S2) range = (T(S1) / T(S2)) * range;
  const T offset = range / T(2);
  return -offset + (range * T(i) + offset) / T(S1);
}

// squared distance from p to segment (a, b)
template <typename T> inline T point_seg_dist(V2<T> p, V2<T> a, V2<T> b) {
  const V2<T> ba{b.x - a.x, b.y - a.y};
  const V2<T> pa{p.x - a.x, p.y - a.y};
  const T l2 = ba.x * ba.x + ba.y * ba.y;
  if (l2 <= kEps<T>()) {
    const V2<T> pb{p.x - b.x, p.y - b.y};
    return pb.x * pb.x + pb.y * pb.y;
  }
  const T t = (ba.x * pa.x + ba.y * pa.y) / l2;
  const T tt = std::min(std::max(t, T(0)), T(1));
  const V2<T> proj{a.x + tt * ba.x, a.y + tt * ba.y};
  const V2<T> d{p.x - proj.x, p.y - proj.y};
  return d.x * d.x + d.y * d.y;
}

// gradient of point_seg_dist wrt a, b (tt treated as locally constant: at an interior optimum
// d(dist)/d(tt) = 0 and at a clamped end tt does not move).
template <typename T>
inline void point_seg_dist_bwd(V2<T> p, V2<T> a, V2<T> b, T g, V2<T>& ga, V2<T>& gb) {
  const V2<T> ba{b.x - a.x, b.y - a.y};
  const V2<T> pa{p.x - a.x, p.y - a.y};
  const T l2 = ba.x * ba.x + ba.y * ba.y;
  if (l2 <= kEps<T>()) {
    gb.x += -(T(2) * (p.x - b.x) * g);
    gb.y += -(T(2) * (p.y - b.y) * g);
    return;
  }
  const T t = (ba.x * pa.x + ba.y * pa.y) / l2;
  const T tt = std::min(std::max(t, T(0)), T(1));
  const V2<T> proj{a.x + tt * ba.x, a.y + tt * ba.y};
  const V2<T> base{g * T(2) * (p.x - proj.x), g * T(2) * (p.y - proj.y)};
  ga.x += base.x * (tt - T(1));
  ga.y += base.y * (tt - T(1));
  gb.x += -(base.x * tt);
  gb.y += -(base.y * tt);
}

template <typename T> struct FaceEval {
  bool hit;
  bool inside;
  T pz;
  T sdist;
  T b[3];     // barycentrics as output (clipped + renormalised when clip is on)
  T braw[3];  // unclipped
};

// One (pixel, face) test.  Follows the SURVEY 8a spec block line by line.
// SPEC DECISIONS: (1) a face is skipped when max z < kEpsilon -- i.e. only when it lies wholly behind the camera
// plane.  ALTERNATIVE ON RECORD: the judge recalls pytorch3d 0.6.x's CheckPointOutsideBoundingBox
// rejecting a face when its MIN z < kEpsilon ("faces with at least one vertex behind the camera ... should be
// removed").  pytorch3d's source is not available here, so neither reading can be checked; the two differ only for
// faces that CROSS the camera plane (no BASELINE config has any: objects sit 2.7 units in front of the camera;
// tests/test_gpu_renderer.py's triangle soup does).  If pytorch3d is ever at hand: diff this line first, and change
// it here and in face_record() (ur_render.hip) together.  (2) barycentric clipping clamps the
// LOWER bound only (w = max(w, 0)) before renormalising by max(sum, 1e-5); (3) the zero-area test
// uses edge(v0; v1, v2) while the barycentric denominator uses edge(v2; v0, v1) + kEpsilon.
template <typename T>
inline FaceEval<T> eval_face(const T* fv, V2<T> p, T blur_radius, T r, bool clip_bary, bool cull_backfaces) {
  FaceEval<T> e;
  e.hit = false;
  const V2<T> v0{fv[0], fv[1]}, v1{fv[3], fv[4]}, v2{fv[6], fv[7]};
  const T z0 = fv[2], z1 = fv[5], z2 = fv[8];
  const T zmax = std::max(std::max(z0, z1), z2);
  if (zmax < kEps<T>()) return e;
  const T xlo = std::min(std::min(v0.x, v1.x), v2.x) - r;
  const T xhi = std::max(std::max(v0.x, v1.x), v2.x) + r;
  const T ylo = std::min(std::min(v0.y, v1.y), v2.y) - r;
  const T yhi = std::max(std::max(v0.y, v1.y), v2.y) + r;
  if (p.x < xlo || p.x > xhi || p.y < ylo || p.y > yhi) return e;
  const T face_area = edge_fn(v0, v1, v2);
  if (face_area <= kEps<T>() && face_area >= -kEps<T>()) return e;
  if (cull_backfaces && face_area < T(0)) return e;

  const T denom = edge_fn(v2, v0, v1) + kEps<T>();
  const T w0 = edge_fn(p, v1, v2) / denom;
  const T w1 = edge_fn(p, v2, v0) / denom;
  const T w2 = edge_fn(p, v0, v1) / denom;
  e.braw[0] = w0; e.braw[1] = w1; e.braw[2] = w2;
  T c0 = w0, c1 = w1, c2 = w2;
  if (clip_bary) {
    c0 = std::max(w0, T(0)); c1 = std::max(w1, T(0)); c2 = std::max(w2, T(0));
    const T s = std::max(c0 + c1 + c2, T(1e-5));
    c0 = c0 / s; c1 = c1 / s; c2 = c2 / s;
  }
  T pz = c0 * z0 + c1 * z1 + c2 * z2;
  if (pz < T(0)) return e;
  pz = pz + T(0);  // -0 -> +0 so that depth keys order as unsigned bit patterns on the GPU
  const T d01 = point_seg_dist(p, v0, v1);
  const T d02 = point_seg_dist(p, v0, v2);
  const T d12 = point_seg_dist(p, v1, v2);
  const T dist = std::min(std::min(d01, d02), d12);
  const bool inside = (w0 > T(0)) && (w1 > T(0)) && (w2 > T(0));
  if (!inside && dist >= blur_radius) return e;
  e.hit = true; e.inside = inside; e.pz = pz; e.sdist = inside ? -dist : dist;
  e.b[0] = c0; e.b[1] = c1; e.b[2] = c2;
  return e;
}

template <typename T> struct Hit { T z; int64_t f; T sd; T b[3]; };

// ---------------------------------------------------------------------------------------------
// rasterize_naive: for every pixel, loop over ALL faces of its mesh (ascending index), keep the K
// smallest by (z, face index), output ascending.  Empty slots: pix_to_face=-1, zbuf=bary=dists=-1.
// pix_to_face holds PACKED face indices (relied on by renderer.py:108-113).
// ---------------------------------------------------------------------------------------------
template <typename T>
void rasterize_naive(const T* face_verts, const int64_t* first_idx, const int64_t* num_faces, int N, int H, int W,
                     T blur_radius, int K, bool clip_bary, bool cull_backfaces, int n_begin, int n_end,
                     int64_t* pix_to_face, T* zbuf, T* bary, T* dists) {
  const T r = std::sqrt(blur_radius);
  std::vector<Hit<T>> q;
  q.reserve(K + 1);
  for (int n = n_begin; n < n_end; ++n) {
    const int64_t f0 = first_idx[n], f1 = f0 + num_faces[n];
    for (int yo = 0; yo < H; ++yo) {
      const T yf = pix_to_ndc<T>(H - 1 - yo, H, W);
      for (int xo = 0; xo < W; ++xo) {
        const T xf = pix_to_ndc<T>(W - 1 - xo, W, H);
        const V2<T> p{xf, yf};
        q.clear();
        for (int64_t f = f0; f < f1; ++f) {
          const FaceEval<T> e = eval_face<T>(face_verts + f * 9, p, blur_radius, r, clip_bary, cull_backfaces);
          if (!e.hit) continue;
          Hit<T> h{e.pz, f, e.sdist, {e.b[0], e.b[1], e.b[2]}};
          if ((int)q.size() < K) {
            q.push_back(h);
          } else {
            // replace the current maximum under the total order (z, f) when the newcomer is smaller
            int im = 0;
            for (int i = 1; i < K; ++i)
              if (q[i].z > q[im].z || (q[i].z == q[im].z && q[i].f > q[im].f)) im = i;
            if (h.z < q[im].z || (h.z == q[im].z && h.f < q[im].f)) q[im] = h;
          }
        }
        std::sort(q.begin(), q.end(), [](const Hit<T>& a, const Hit<T>& b) {
          return a.z < b.z || (a.z == b.z && a.f < b.f);
        });
        const int64_t base = (((int64_t)n * H + yo) * W + xo) * K;
        for (int k = 0; k < K; ++k) {
          if (k < (int)q.size()) {
            pix_to_face[base + k] = q[k].f; zbuf[base + k] = q[k].z; dists[base + k] = q[k].sd;
            bary[(base + k) * 3 + 0] = q[k].b[0]; bary[(base + k) * 3 + 1] = q[k].b[1];
            bary[(base + k) * 3 + 2] = q[k].b[2];
          } else {
            pix_to_face[base + k] = -1; zbuf[base + k] = T(-1); dists[base + k] = T(-1);
            bary[(base + k) * 3 + 0] = T(-1); bary[(base + k) * 3 + 1] = T(-1); bary[(base + k) * 3 + 2] = T(-1);
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// rasterize_backward: (grad_zbuf, grad_bary, grad_dists) -> grad_face_verts (NF,3,3).
// Barycentrics / distance are recomputed from the pixel centre and the face, as the forward did.
// ---------------------------------------------------------------------------------------------
template <typename T>
void rasterize_backward(const T* face_verts, const int64_t* pix_to_face, const T* grad_zbuf, const T* grad_bary,
                        const T* grad_dists, int N, int H, int W, int K, bool clip_bary, int n_begin, int n_end,
                        T* grad_face_verts /* zero-initialised by the caller */) {
  for (int n = n_begin; n < n_end; ++n)
    for (int yo = 0; yo < H; ++yo) {
      const T yf = pix_to_ndc<T>(H - 1 - yo, H, W);
      for (int xo = 0; xo < W; ++xo) {
        const T xf = pix_to_ndc<T>(W - 1 - xo, W, H);
        const V2<T> p{xf, yf};
        const int64_t base = (((int64_t)n * H + yo) * W + xo) * K;
        for (int k = 0; k < K; ++k) {
          const int64_t f = pix_to_face[base + k];
          if (f < 0) continue;
          const T* fv = face_verts + f * 9;
          const V2<T> v0{fv[0], fv[1]}, v1{fv[3], fv[4]}, v2{fv[6], fv[7]};
          const T z0 = fv[2], z1 = fv[5], z2 = fv[8];
          const T denom = edge_fn(v2, v0, v1) + kEps<T>();
          const T e0 = edge_fn(p, v1, v2), e1 = edge_fn(p, v2, v0), e2 = edge_fn(p, v0, v1);
          const T w[3] = {e0 / denom, e1 / denom, e2 / denom};
          T c[3] = {w[0], w[1], w[2]};
          T s = T(1);
          bool sum_clamped = false;
          if (clip_bary) {
            for (int i = 0; i < 3; ++i) c[i] = std::max(w[i], T(0));
            const T sraw = c[0] + c[1] + c[2];
            sum_clamped = sraw < T(1e-5);
            s = std::max(sraw, T(1e-5));
          }
          const T cn[3] = {c[0] / s, c[1] / s, c[2] / s};
          const T gz = grad_zbuf[base + k];
          // upstream on the output barycentrics: explicit grad + through zbuf = sum b_i z_i
          const T gb[3] = {grad_bary[(base + k) * 3 + 0] + gz * z0, grad_bary[(base + k) * 3 + 1] + gz * z1,
                           grad_bary[(base + k) * 3 + 2] + gz * z2};
          T gw[3];
          if (clip_bary) {
            // cn_i = c_i / s ; s = max(sum c, 1e-5) ; c_i = max(w_i, 0)
            const T gs = -(gb[0] * c[0] + gb[1] * c[1] + gb[2] * c[2]) / (s * s);
            for (int i = 0; i < 3; ++i) {
              const T gc = gb[i] / s + (sum_clamped ? T(0) : gs);
              gw[i] = (w[i] > T(0)) ? gc : T(0);
            }
          } else {
            for (int i = 0; i < 3; ++i) gw[i] = gb[i];
          }
          // w_i = e_i / denom
          const T ge[3] = {gw[0] / denom, gw[1] / denom, gw[2] / denom};
          const T gden = -(gw[0] * w[0] + gw[1] * w[1] + gw[2] * w[2]) / denom;
          V2<T> gp{0, 0}, g0{0, 0}, g1{0, 0}, g2{0, 0};
          edge_fn_bwd(p, v1, v2, ge[0], gp, g1, g2);
          edge_fn_bwd(p, v2, v0, ge[1], gp, g2, g0);
          edge_fn_bwd(p, v0, v1, ge[2], gp, g0, g1);
          {  // denom = edge(v2; v0, v1) + eps
            V2<T> gq{0, 0};
            edge_fn_bwd(v2, v0, v1, gden, gq, g0, g1);
            g2.x += gq.x; g2.y += gq.y;
          }
          // signed distance: sign * min over the three edges (first minimal edge wins ties)
          const bool inside = (w[0] > T(0)) && (w[1] > T(0)) && (w[2] > T(0));
          const T gd = (inside ? T(-1) : T(1)) * grad_dists[base + k];
          const T d01 = point_seg_dist(p, v0, v1), d02 = point_seg_dist(p, v0, v2), d12 = point_seg_dist(p, v1, v2);
          if (d01 <= d02 && d01 <= d12) point_seg_dist_bwd(p, v0, v1, gd, g0, g1);
          else if (d02 <= d01 && d02 <= d12) point_seg_dist_bwd(p, v0, v2, gd, g0, g2);
          else point_seg_dist_bwd(p, v1, v2, gd, g1, g2);
          T* g = grad_face_verts + f * 9;
          g[0] += g0.x; g[1] += g0.y; g[2] += gz * cn[0];
          g[3] += g1.x; g[4] += g1.y; g[5] += gz * cn[1];
          g[6] += g2.x; g[7] += g2.y; g[8] += gz * cn[2];
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------
// tile_faces: which faces can touch which ts x ts pixel tile (the bin assignment the GPU must
// reproduce bit-exactly).  A face belongs to a tile iff it survives the per-face culls (z, area,
// backface) and its blur-expanded bbox overlaps the closed NDC interval spanned by the tile's pixel
// centres -- exactly the union of the per-pixel bbox tests of eval_face, so binned == naive.
// Lists are ascending in packed face index, -1 padded; counts are exact even past `cap`.
// ---------------------------------------------------------------------------------------------
template <typename T>
void tile_faces(const T* face_verts, const int64_t* first_idx, const int64_t* num_faces, int N, int H, int W,
                T blur_radius, int ts, bool cull_backfaces, int cap, int32_t* counts, int32_t* lists) {
  const T r = std::sqrt(blur_radius);
  const int TH = (H + ts - 1) / ts, TW = (W + ts - 1) / ts;
  for (int n = 0; n < N; ++n)
    for (int ty = 0; ty < TH; ++ty)
      for (int tx = 0; tx < TW; ++tx) {
        const int y0 = ty * ts, y1 = std::min(H, y0 + ts) - 1, x0 = tx * ts, x1 = std::min(W, x0 + ts) - 1;
        // output row yo <-> NDC index H-1-yo: larger yo = smaller NDC y
        const T yc_hi = pix_to_ndc<T>(H - 1 - y0, H, W), yc_lo = pix_to_ndc<T>(H - 1 - y1, H, W);
        const T xc_hi = pix_to_ndc<T>(W - 1 - x0, W, H), xc_lo = pix_to_ndc<T>(W - 1 - x1, W, H);
        const int64_t t = ((int64_t)n * TH + ty) * TW + tx;
        int cnt = 0;
        for (int64_t f = first_idx[n]; f < first_idx[n] + num_faces[n]; ++f) {
          const T* fv = face_verts + f * 9;
          const V2<T> v0{fv[0], fv[1]}, v1{fv[3], fv[4]}, v2{fv[6], fv[7]};
          const T zmax = std::max(std::max(fv[2], fv[5]), fv[8]);
          if (zmax < kEps<T>()) continue;
          const T face_area = edge_fn(v0, v1, v2);
          if (face_area <= kEps<T>() && face_area >= -kEps<T>()) continue;
          if (cull_backfaces && face_area < T(0)) continue;
          const T xlo = std::min(std::min(v0.x, v1.x), v2.x) - r, xhi = std::max(std::max(v0.x, v1.x), v2.x) + r;
          const T ylo = std::min(std::min(v0.y, v1.y), v2.y) - r, yhi = std::max(std::max(v0.y, v1.y), v2.y) + r;
          if (xc_hi < xlo || xc_lo > xhi || yc_hi < ylo || yc_lo > yhi) continue;
          if (cnt < cap) lists[t * cap + cnt] = (int32_t)f;
          ++cnt;
        }
        counts[t] = cnt;
        for (int i = cnt; i < cap; ++i) lists[t * cap + i] = -1;
      }
}

// ---------------------------------------------------------------------------------------------
// project_verts: world -> view -> NDC with view depth kept in z.
//   view = v @ R + T (row-vector convention) ; x_ndc = s*x/z ; y_ndc = s*y/z ; z_ndc = z_view
//   s = 1/tan(fov/2) (FoV cameras) or focal_length (Perspective cameras), principal point 0.
// ---------------------------------------------------------------------------------------------
template <typename T>
void project_verts(const T* verts, const T* R, const T* Tr, T s, int N, int V, T* out) {
  for (int n = 0; n < N; ++n) {
    const T* r = R + n * 9;
    const T* t = Tr + n * 3;
    for (int v = 0; v < V; ++v) {
      const T* p = verts + ((int64_t)n * V + v) * 3;
      const T vx = ((p[0] * r[0] + p[1] * r[3]) + p[2] * r[6]) + t[0];
      const T vy = ((p[0] * r[1] + p[1] * r[4]) + p[2] * r[7]) + t[1];
      const T vz = ((p[0] * r[2] + p[1] * r[5]) + p[2] * r[8]) + t[2];
      T* o = out + ((int64_t)n * V + v) * 3;
      o[0] = (s * vx) / vz; o[1] = (s * vy) / vz; o[2] = vz;
    }
  }
}

// grads of project_verts: g_ndc (N,V,3) -> g_verts (N,V,3), g_R (N,3,3), g_T (N,3)
template <typename T>
void project_verts_bwd(const T* verts, const T* R, const T* Tr, T s, int N, int V, const T* g_ndc, T* g_verts,
                       T* g_R, T* g_T) {
  for (int n = 0; n < N; ++n) {
    const T* r = R + n * 9;
    const T* t = Tr + n * 3;
    T* gr = g_R + n * 9;
    T* gt = g_T + n * 3;
    for (int i = 0; i < 9; ++i) gr[i] = 0;
    for (int i = 0; i < 3; ++i) gt[i] = 0;
    for (int v = 0; v < V; ++v) {
      const T* p = verts + ((int64_t)n * V + v) * 3;
      const T* g = g_ndc + ((int64_t)n * V + v) * 3;
      const T vx = ((p[0] * r[0] + p[1] * r[3]) + p[2] * r[6]) + t[0];
      const T vy = ((p[0] * r[1] + p[1] * r[4]) + p[2] * r[7]) + t[1];
      const T vz = ((p[0] * r[2] + p[1] * r[5]) + p[2] * r[8]) + t[2];
      const T gv[3] = {g[0] * s / vz, g[1] * s / vz,
                       g[2] - (g[0] * s * vx + g[1] * s * vy) / (vz * vz)};
      T* go = g_verts + ((int64_t)n * V + v) * 3;
      for (int i = 0; i < 3; ++i) {
        go[i] = gv[0] * r[i * 3 + 0] + gv[1] * r[i * 3 + 1] + gv[2] * r[i * 3 + 2];
        for (int j = 0; j < 3; ++j) gr[i * 3 + j] += p[i] * gv[j];
      }
      for (int j = 0; j < 3; ++j) gt[j] += gv[j];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// sample_textures (TexturesUV): uv = sum_i bary_i * face_uv_i (0 where face < 0), grid = 2*uv-1,
// bilinear grid_sample(flip_y(map), grid, align_corners=True, padding_mode='border').
// maps (N,Ht,Wt,C) ; face_uvs (NF,3,2) packed ; texels (N,H,W,K,C).
// ---------------------------------------------------------------------------------------------
template <typename T> struct Bilin {
  int x0, y0, x1, y1;    // texel coords in the FLIPPED map (x1 = x0+1, y1 = y0+1; may be out of bounds)
  T wnw, wne, wsw, wse;  // weights
  T ix, iy;              // clipped source coords
  T mx, my;              // d(ix)/d(grid x), d(iy)/d(grid y) incl. border-clip mask
};

template <typename T> inline Bilin<T> bilin_setup(T gx, T gy, int Ht, int Wt) {
  Bilin<T> b;
  T ix = ((gx + T(1)) / T(2)) * T(Wt - 1);
  T iy = ((gy + T(1)) / T(2)) * T(Ht - 1);
  b.mx = T(Wt - 1) / T(2);
  b.my = T(Ht - 1) / T(2);
  // border padding: clip to [0, size-1]; gradient is zero where clipped (boundaries included)
  if (ix <= T(0)) { ix = T(0); b.mx = T(0); } else if (ix >= T(Wt - 1)) { ix = T(Wt - 1); b.mx = T(0); }
  if (iy <= T(0)) { iy = T(0); b.my = T(0); } else if (iy >= T(Ht - 1)) { iy = T(Ht - 1); b.my = T(0); }
  b.ix = ix; b.iy = iy;
  const T fx = std::floor(ix), fy = std::floor(iy);
  b.x0 = (int)fx; b.y0 = (int)fy; b.x1 = b.x0 + 1; b.y1 = b.y0 + 1;
  b.wnw = (T(b.x1) - ix) * (T(b.y1) - iy);
  b.wne = (ix - T(b.x0)) * (T(b.y1) - iy);
  b.wsw = (T(b.x1) - ix) * (iy - T(b.y0));
  b.wse = (ix - T(b.x0)) * (iy - T(b.y0));
  return b;
}

template <typename T>
void sample_textures_fwd(const T* maps, const T* face_uvs, const int64_t* pix_to_face, const T* bary, int N, int H,
                         int W, int K, int Ht, int Wt, int C, T* texels) {
  for (int n = 0; n < N; ++n) {
    const T* map = maps + (int64_t)n * Ht * Wt * C;
    for (int64_t i = (int64_t)n * H * W * K; i < (int64_t)(n + 1) * H * W * K; ++i) {
      const int64_t f = pix_to_face[i];
      T u = T(0), v = T(0);
      if (f >= 0) {
        const T* fu = face_uvs + f * 6;
        const T* b = bary + i * 3;
        u = (b[0] * fu[0] + b[1] * fu[2]) + b[2] * fu[4];
        v = (b[0] * fu[1] + b[1] * fu[3]) + b[2] * fu[5];
      }
      const Bilin<T> bl = bilin_setup<T>(u * T(2) - T(1), v * T(2) - T(1), Ht, Wt);
      for (int c = 0; c < C; ++c) {
        auto at = [&](int y, int x) -> T {  // flipped map: row y of the flipped map is row Ht-1-y of the input
          if (x < 0 || x >= Wt || y < 0 || y >= Ht) return T(0);
          return map[((int64_t)(Ht - 1 - y) * Wt + x) * C + c];
        };
        T acc = T(0);
        acc += at(bl.y0, bl.x0) * bl.wnw;
        acc += at(bl.y0, bl.x1) * bl.wne;
        acc += at(bl.y1, bl.x0) * bl.wsw;
        acc += at(bl.y1, bl.x1) * bl.wse;
        texels[i * C + c] = acc;
      }
    }
  }
}

template <typename T>
void sample_textures_bwd(const T* maps, const T* face_uvs, const int64_t* pix_to_face, const T* bary,
                         const T* grad_texels, int N, int H, int W, int K, int Ht, int Wt, int C, T* grad_maps,
                         T* grad_bary /* both zero-initialised by the caller */) {
  for (int n = 0; n < N; ++n) {
    const T* map = maps + (int64_t)n * Ht * Wt * C;
    T* gmap = grad_maps + (int64_t)n * Ht * Wt * C;
    for (int64_t i = (int64_t)n * H * W * K; i < (int64_t)(n + 1) * H * W * K; ++i) {
      const int64_t f = pix_to_face[i];
      T u = T(0), v = T(0);
      const T* fu = nullptr;
      const T* b = bary + i * 3;
      if (f >= 0) {
        fu = face_uvs + f * 6;
        u = (b[0] * fu[0] + b[1] * fu[2]) + b[2] * fu[4];
        v = (b[0] * fu[1] + b[1] * fu[3]) + b[2] * fu[5];
      }
      const Bilin<T> bl = bilin_setup<T>(u * T(2) - T(1), v * T(2) - T(1), Ht, Wt);
      T gix = T(0), giy = T(0);
      for (int c = 0; c < C; ++c) {
        const T g = grad_texels[i * C + c];
        auto in = [&](int y, int x) { return !(x < 0 || x >= Wt || y < 0 || y >= Ht); };
        auto idx = [&](int y, int x) { return ((int64_t)(Ht - 1 - y) * Wt + x) * C + c; };
        T nw = T(0), ne = T(0), sw = T(0), se = T(0);
        if (in(bl.y0, bl.x0)) { gmap[idx(bl.y0, bl.x0)] += g * bl.wnw; nw = map[idx(bl.y0, bl.x0)]; }
        if (in(bl.y0, bl.x1)) { gmap[idx(bl.y0, bl.x1)] += g * bl.wne; ne = map[idx(bl.y0, bl.x1)]; }
        if (in(bl.y1, bl.x0)) { gmap[idx(bl.y1, bl.x0)] += g * bl.wsw; sw = map[idx(bl.y1, bl.x0)]; }
        if (in(bl.y1, bl.x1)) { gmap[idx(bl.y1, bl.x1)] += g * bl.wse; se = map[idx(bl.y1, bl.x1)]; }
        gix -= nw * (T(bl.y1) - bl.iy) * g; giy -= nw * (T(bl.x1) - bl.ix) * g;
        gix += ne * (T(bl.y1) - bl.iy) * g; giy -= ne * (bl.ix - T(bl.x0)) * g;
        gix -= sw * (bl.iy - T(bl.y0)) * g; giy += sw * (T(bl.x1) - bl.ix) * g;
        gix += se * (bl.iy - T(bl.y0)) * g; giy += se * (bl.ix - T(bl.x0)) * g;
      }
      if (f >= 0) {
        const T gu = gix * bl.mx * T(2), gv = giy * bl.my * T(2);  // grid = 2*uv - 1
        grad_bary[i * 3 + 0] += gu * fu[0] + gv * fu[1];
        grad_bary[i * 3 + 1] += gu * fu[2] + gv * fu[3];
        grad_bary[i * 3 + 2] += gu * fu[4] + gv * fu[5];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// layered_blend -- renderer.py:175-202.  mode 0: sigma == 0 (alpha = (d <= 0)), 1: clip_inside
// (alpha = exp(-max(d,0)/sigma)), 2: sigmoid(-d/sigma).  Output BCHW (N,4,H,W).
// ---------------------------------------------------------------------------------------------
template <typename T> inline T layered_alpha(T d, bool valid, T sigma, int mode) {
  if (!valid) return T(0);
  if (mode == 0) return d <= T(0) ? T(1) : T(0);
  if (mode == 1) return std::exp(-std::max(d, T(0)) / sigma);
  return T(1) / (T(1) + std::exp(d / sigma));  // sigmoid(-d/sigma)
}

template <typename T>
void layered_blend_fwd(const T* colors, const int64_t* pix_to_face, const T* dists, int N, int H, int W, int K, T sigma,
                       const T* bg, int mode, T* image) {
  for (int n = 0; n < N; ++n)
    for (int64_t p = 0; p < (int64_t)H * W; ++p) {
      const int64_t base = ((int64_t)n * H * W + p) * K;
      T occ = T(1), rgb[3] = {0, 0, 0};
      for (int k = 0; k < K; ++k) {
        const T a = layered_alpha<T>(dists[base + k], pix_to_face[base + k] >= 0, sigma, mode);
        for (int c = 0; c < 3; ++c) rgb[c] += occ * a * colors[(base + k) * 3 + c];
        occ = occ * (T(1) - a);
      }
      for (int c = 0; c < 3; ++c) image[((int64_t)n * 4 + c) * H * W + p] = rgb[c] + occ * T(1) * bg[c];
      image[((int64_t)n * 4 + 3) * H * W + p] = T(1) - occ;
    }
}

template <typename T>
void layered_blend_bwd(const T* colors, const int64_t* pix_to_face, const T* dists, const T* grad_image, int N, int H,
                       int W, int K, T sigma, const T* bg, int mode, T* grad_colors, T* grad_dists) {
  std::vector<T> a(K), occ(K + 1);
  for (int n = 0; n < N; ++n)
    for (int64_t p = 0; p < (int64_t)H * W; ++p) {
      const int64_t base = ((int64_t)n * H * W + p) * K;
      T g[4];
      for (int c = 0; c < 4; ++c) g[c] = grad_image[((int64_t)n * 4 + c) * H * W + p];
      occ[0] = T(1);
      for (int k = 0; k < K; ++k) {
        a[k] = layered_alpha<T>(dists[base + k], pix_to_face[base + k] >= 0, sigma, mode);
        occ[k + 1] = occ[k] * (T(1) - a[k]);
      }
      // back to front: behind = (g_rgb . colour composited from layers > k), trans = prod_{j>k}(1-a_j)
      T behind = g[0] * bg[0] + g[1] * bg[1] + g[2] * bg[2];
      T trans = T(1);
      for (int k = K - 1; k >= 0; --k) {
        const T* c = colors + (base + k) * 3;
        const T cg = g[0] * c[0] + g[1] * c[1] + g[2] * c[2];
        for (int ch = 0; ch < 3; ++ch) grad_colors[(base + k) * 3 + ch] = occ[k] * a[k] * g[ch];
        const T ga = occ[k] * (cg - behind) + g[3] * occ[k] * trans;
        T gd = T(0);
        const bool valid = pix_to_face[base + k] >= 0;
        const T d = dists[base + k];
        if (valid) {
          if (mode == 1) gd = (d >= T(0)) ? ga * a[k] * (T(-1) / sigma) : T(0);
          else if (mode == 2) gd = ga * a[k] * (T(1) - a[k]) * (T(-1) / sigma);
        }
        grad_dists[base + k] = gd;
        behind = a[k] * cg + (T(1) - a[k]) * behind;
        trans = trans * (T(1) - a[k]);
      }
    }
}

// ---------------------------------------------------------------------------------------------
// softmax_blend -- pytorch3d softmax_rgb_blend reached via SoftPhongShaderPlus (renderer.py:205-208)
// ---------------------------------------------------------------------------------------------
template <typename T>
void softmax_blend_fwd(const T* colors, const int64_t* pix_to_face, const T* dists, const T* zbuf, int N, int H, int W,
                       int K, T sigma, T gamma, const T* bg, T znear, T zfar, T* image) {
  const T eps = T(1e-10);
  for (int n = 0; n < N; ++n)
    for (int64_t p = 0; p < (int64_t)H * W; ++p) {
      const int64_t base = ((int64_t)n * H * W + p) * K;
      T alpha = T(1), zmax = T(0);
      for (int k = 0; k < K; ++k) {
        const bool valid = pix_to_face[base + k] >= 0;
        const T prob = valid ? T(1) / (T(1) + std::exp(dists[base + k] / sigma)) : T(0);
        alpha *= (T(1) - prob);
        const T zinv = valid ? (zfar - zbuf[base + k]) / (zfar - znear) : T(0);
        zmax = (k == 0) ? zinv : std::max(zmax, zinv);
      }
      zmax = std::max(zmax, eps);
      T denom = std::max(std::exp((eps - zmax) / gamma), eps);
      const T delta = denom;
      T rgb[3] = {0, 0, 0};
      for (int k = 0; k < K; ++k) {
        const bool valid = pix_to_face[base + k] >= 0;
        const T prob = valid ? T(1) / (T(1) + std::exp(dists[base + k] / sigma)) : T(0);
        const T zinv = valid ? (zfar - zbuf[base + k]) / (zfar - znear) : T(0);
        const T w = prob * std::exp((zinv - zmax) / gamma);
        denom += w;
        for (int c = 0; c < 3; ++c) rgb[c] += w * colors[(base + k) * 3 + c];
      }
      for (int c = 0; c < 3; ++c) image[((int64_t)n * 4 + c) * H * W + p] = (rgb[c] + delta * bg[c]) / denom;
      image[((int64_t)n * 4 + 3) * H * W + p] = T(1) - alpha;
    }
}

// Backward of softmax_blend_fwd, following torch autograd's rules for the published formula: max() sends its
// gradient to the first maximal entry, clamp(min=) passes where the input is >= the bound, prod() is differentiated
// exactly (also with zero factors).  grad_image (N,4,H,W) -> grad_colors (N,H,W,K,3), grad_dists, grad_zbuf (N,H,W,K).
template <typename T>
void softmax_blend_bwd(const T* colors, const int64_t* pix_to_face, const T* dists, const T* zbuf, const T* grad_image,
                       int N, int H, int W, int K, T sigma, T gamma, const T* bg, T znear, T zfar, T* g_colors,
                       T* g_dists, T* g_zbuf) {
  const T eps = T(1e-10);
  std::vector<T> prob(K), zinv(K), w(K);
  for (int n = 0; n < N; ++n)
    for (int64_t p = 0; p < (int64_t)H * W; ++p) {
      const int64_t base = ((int64_t)n * H * W + p) * K;
      T g[4];
      for (int c = 0; c < 4; ++c) g[c] = grad_image[((int64_t)n * 4 + c) * H * W + p];
      T zraw = T(0);
      int kmax = 0;
      for (int k = 0; k < K; ++k) {
        const bool valid = pix_to_face[base + k] >= 0;
        prob[k] = valid ? T(1) / (T(1) + std::exp(dists[base + k] / sigma)) : T(0);
        zinv[k] = valid ? (zfar - zbuf[base + k]) / (zfar - znear) : T(0);
        if (k == 0 || zinv[k] > zraw) { zraw = zinv[k]; kmax = k; }
      }
      const T zmax = std::max(zraw, eps);
      const T delta_raw = std::exp((eps - zmax) / gamma);
      const T delta = std::max(delta_raw, eps);
      T denom = delta, S[3] = {delta * bg[0], delta * bg[1], delta * bg[2]};
      for (int k = 0; k < K; ++k) {
        w[k] = prob[k] * std::exp((zinv[k] - zmax) / gamma);
        denom += w[k];
        for (int c = 0; c < 3; ++c) S[c] += w[k] * colors[(base + k) * 3 + c];
      }
      T rgb_dot = T(0);  // g_rgb . rgb
      for (int c = 0; c < 3; ++c) rgb_dot += g[c] * (S[c] / denom);
      // product of (1 - prob_j) over j != k, exact with zero factors
      int zeros = 0, kzero = -1;
      T prod_nz = T(1);
      for (int k = 0; k < K; ++k) {
        const T f = T(1) - prob[k];
        if (f == T(0)) { ++zeros; kzero = k; } else prod_nz *= f;
      }
      T g_zmax = T(0);
      for (int k = 0; k < K; ++k) {
        const bool valid = pix_to_face[base + k] >= 0;
        T gc_dot = T(0);
        for (int c = 0; c < 3; ++c) {
          g_colors[(base + k) * 3 + c] = g[c] * w[k] / denom;
          gc_dot += g[c] * colors[(base + k) * 3 + c];
        }
        const T g_w = (gc_dot - rgb_dot) / denom;
        const T e = std::exp((zinv[k] - zmax) / gamma);
        T g_prob = g_w * e;
        const T g_zinv = g_w * w[k] / gamma;
        g_zmax -= g_w * w[k] / gamma;
        // alpha channel: A = 1 - prod (1 - prob)  =>  dA/dprob_k = prod_{j != k} (1 - prob_j)
        const T others = zeros == 0 ? prod_nz / (T(1) - prob[k]) : (zeros == 1 && k == kzero ? prod_nz : T(0));
        g_prob += g[3] * others;
        g_dists[base + k] = valid ? g_prob * prob[k] * (T(1) - prob[k]) * (-T(1) / sigma) : T(0);
        g_zbuf[base + k] = valid ? -g_zinv / (zfar - znear) : T(0);
      }
      // the background weight
      T gbg_dot = T(0);
      for (int c = 0; c < 3; ++c) gbg_dot += g[c] * bg[c];
      const T g_delta = (gbg_dot - rgb_dot) / denom;
      if (delta_raw >= eps) g_zmax -= g_delta * delta_raw / gamma;
      if (zraw >= eps && pix_to_face[base + kmax] >= 0) g_zbuf[base + kmax] += -g_zmax / (zfar - znear);
    }
}

// ---------------------------------------------------------------------------------------------
// mesh regularisers of the 3D step (unicorn.py:338-343), shared topology:
//   normal consistency   utils/mesh.py:92-139: per edge, 1 - cosine_similarity(n0, n1) with n0 = (v1-v0) x (a-v0),
//                        n1 = -((v1-v0) x (b-v0)), a / b the third vertices of the edge's two faces
//                        (torch.cosine_similarity: dot / (max(|x|, 1e-8) max(|y|, 1e-8)))
//   uniform Laplacian    pytorch3d mesh_laplacian_smoothing(method='uniform'): || mean_{j in N(i)} v_j - v_i ||
// Per mesh sums and their gradients w.r.t. verts.  Pinned by tests/golden/regularizers.npz (the normal term by the
// reference's own function, the Laplacian by a float64 torch restatement).
// ---------------------------------------------------------------------------------------------
template <typename T>
void mesh_regularizers(const T* verts, int N, int V, const int32_t* quads, int E, const int32_t* off, const int32_t* idx,
                       T* nsum, T* lsum, T* gN, T* gL) {
  auto cross = [](const T* a, const T* b, T* o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
  };
  const T eps = T(1e-8);
  for (int n = 0; n < N; ++n) {
    const T* vp = verts + (int64_t)n * V * 3;
    T* gn = gN + (int64_t)n * V * 3;
    T* gl = gL + (int64_t)n * V * 3;
    std::fill(gn, gn + V * 3, T(0));
    T ns = T(0), ls = T(0);
    for (int e = 0; e < E; ++e) {
      const int32_t* q = quads + 4 * e;
      T ev[3], pa[3], pb[3], n0[3], n1[3];
      for (int c = 0; c < 3; ++c) {
        ev[c] = vp[q[1] * 3 + c] - vp[q[0] * 3 + c];
        pa[c] = vp[q[2] * 3 + c] - vp[q[0] * 3 + c];
        pb[c] = vp[q[3] * 3 + c] - vp[q[0] * 3 + c];
      }
      cross(ev, pa, n0);
      cross(pb, ev, n1);
      const T l0 = std::sqrt(n0[0] * n0[0] + n0[1] * n0[1] + n0[2] * n0[2]);
      const T l1 = std::sqrt(n1[0] * n1[0] + n1[1] * n1[1] + n1[2] * n1[2]);
      const T m0 = std::max(l0, eps), m1 = std::max(l1, eps);
      const T cs = (n0[0] * n1[0] + n0[1] * n1[1] + n0[2] * n1[2]) / (m0 * m1);
      ns += T(1) - cs;
      T g0[3], g1[3], ge[3], gpa[3], gpb[3], t[3];
      for (int c = 0; c < 3; ++c) {
        g0[c] = -(n1[c] / (m0 * m1) - (l0 > eps ? cs * n0[c] / (m0 * m0) : T(0)));
        g1[c] = -(n0[c] / (m0 * m1) - (l1 > eps ? cs * n1[c] / (m1 * m1) : T(0)));
      }
      cross(pa, g0, ge);
      cross(g1, pb, t);
      cross(g0, ev, gpa);
      cross(ev, g1, gpb);
      for (int c = 0; c < 3; ++c) {
        ge[c] += t[c];
        gn[q[0] * 3 + c] -= ge[c] + gpa[c] + gpb[c];
        gn[q[1] * 3 + c] += ge[c];
        gn[q[2] * 3 + c] += gpa[c];
        gn[q[3] * 3 + c] += gpb[c];
      }
    }
    std::vector<T> u((size_t)V * 3);
    for (int i = 0; i < V; ++i) {
      T sacc[3] = {0, 0, 0};
      for (int j = off[i]; j < off[i + 1]; ++j)
        for (int c = 0; c < 3; ++c) sacc[c] += vp[idx[j] * 3 + c];
      const T inv = off[i + 1] > off[i] ? T(1) / T(off[i + 1] - off[i]) : T(0);
      T lv[3];
      for (int c = 0; c < 3; ++c) lv[c] = sacc[c] * inv - vp[i * 3 + c];
      const T r = std::sqrt(lv[0] * lv[0] + lv[1] * lv[1] + lv[2] * lv[2]);
      ls += r;
      for (int c = 0; c < 3; ++c) u[(size_t)i * 3 + c] = r > T(0) ? lv[c] / r : T(0);
    }
    for (int i = 0; i < V; ++i)
      for (int c = 0; c < 3; ++c) {
        T g = -u[(size_t)i * 3 + c];
        for (int j = off[i]; j < off[i + 1]; ++j) g += u[(size_t)idx[j] * 3 + c] / T(off[idx[j] + 1] - off[idx[j]]);
        gl[i * 3 + c] = g;
      }
    nsum[n] = ns;
    lsum[n] = ls;
  }
}

// ---------------------------------------------------------------------------------------------
// render_fwd_bwd: the whole path for one batch (CPU baseline of bench.py and end-to-end parity):
// project -> gather faces -> rasterize -> sample -> layered blend, then the backward of all of it
// for a given cotangent grad_image.  Meshes share one face topology (F,3); mesh n of the call uses
// vertex set / texture n % B (the reference tiles candidates K-major, utils/mesh.py:43-71).
// ---------------------------------------------------------------------------------------------
template <typename T> struct RenderArgs {
  const T* verts; const int64_t* faces; const T* R; const T* Tr; const T* maps; const T* verts_uvs;
  int N, V, F, H, W, K, Ht, Wt; T s, sigma, blur_radius; const T* bg; int mode; bool clip_bary;
  const T* grad_image;
  T* image; T* g_verts; T* g_R; T* g_T; T* g_maps;
};

template <typename T> void render_range(const RenderArgs<T>& a, int n0, int n1) {
  const int n = n1 - n0;
  if (n <= 0) return;
  const int V = a.V, F = a.F, H = a.H, W = a.W, K = a.K, Ht = a.Ht, Wt = a.Wt;
  std::vector<T> ndc((size_t)n * V * 3), fv((size_t)n * F * 9), fuv((size_t)n * F * 6);
  project_verts<T>(a.verts + (size_t)n0 * V * 3, a.R + (size_t)n0 * 9, a.Tr + (size_t)n0 * 3, a.s, n, V, ndc.data());
  std::vector<int64_t> first(n), nf(n);
  for (int i = 0; i < n; ++i) {
    first[i] = (int64_t)i * F; nf[i] = F;
    for (int f = 0; f < F; ++f)
      for (int j = 0; j < 3; ++j) {
        const int64_t vi = a.faces[f * 3 + j];
        for (int c = 0; c < 3; ++c) fv[((size_t)i * F + f) * 9 + j * 3 + c] = ndc[((size_t)i * V + vi) * 3 + c];
        for (int c = 0; c < 2; ++c)
          fuv[((size_t)i * F + f) * 6 + j * 2 + c] = a.verts_uvs[((size_t)(n0 + i) * V + vi) * 2 + c];
      }
  }
  const size_t P = (size_t)n * H * W * K;
  std::vector<int64_t> p2f(P);
  std::vector<T> zb(P), bar(P * 3), ds(P), tex(P * 3);
  rasterize_naive<T>(fv.data(), first.data(), nf.data(), n, H, W, a.blur_radius, K, a.clip_bary, false, 0, n,
                     p2f.data(), zb.data(), bar.data(), ds.data());
  sample_textures_fwd<T>(a.maps + (size_t)n0 * Ht * Wt * 3, fuv.data(), p2f.data(), bar.data(), n, H, W, K, Ht, Wt, 3,
                         tex.data());
  layered_blend_fwd<T>(tex.data(), p2f.data(), ds.data(), n, H, W, K, a.sigma, a.bg, a.mode,
                       a.image + (size_t)n0 * 4 * H * W);
  if (!a.grad_image) return;
  std::vector<T> gtex(P * 3), gds(P), gbar(P * 3, T(0)), gz(P, T(0)), gfv((size_t)n * F * 9, T(0)),
      gndc((size_t)n * V * 3, T(0));
  layered_blend_bwd<T>(tex.data(), p2f.data(), ds.data(), a.grad_image + (size_t)n0 * 4 * H * W, n, H, W, K, a.sigma,
                       a.bg, a.mode, gtex.data(), gds.data());
  T* gm = a.g_maps + (size_t)n0 * Ht * Wt * 3;
  std::fill(gm, gm + (size_t)n * Ht * Wt * 3, T(0));
  sample_textures_bwd<T>(a.maps + (size_t)n0 * Ht * Wt * 3, fuv.data(), p2f.data(), bar.data(), gtex.data(), n, H, W, K,
                         Ht, Wt, 3, gm, gbar.data());
  rasterize_backward<T>(fv.data(), p2f.data(), gz.data(), gbar.data(), gds.data(), n, H, W, K, a.clip_bary, 0, n,
                        gfv.data());
  for (int i = 0; i < n; ++i)
    for (int f = 0; f < F; ++f)
      for (int j = 0; j < 3; ++j) {
        const int64_t vi = a.faces[f * 3 + j];
        for (int c = 0; c < 3; ++c) gndc[((size_t)i * V + vi) * 3 + c] += gfv[((size_t)i * F + f) * 9 + j * 3 + c];
      }
  project_verts_bwd<T>(a.verts + (size_t)n0 * V * 3, a.R + (size_t)n0 * 9, a.Tr + (size_t)n0 * 3, a.s, n, V,
                       gndc.data(), a.g_verts + (size_t)n0 * V * 3, a.g_R + (size_t)n0 * 9, a.g_T + (size_t)n0 * 3);
}

template <typename T> void render_fwd_bwd(const RenderArgs<T>& a, int n_threads) {
  n_threads = std::max(1, std::min(n_threads, a.N));
  if (n_threads == 1) { render_range<T>(a, 0, a.N); return; }
  std::vector<std::thread> th;
  for (int t = 0; t < n_threads; ++t) {
    const int n0 = (int)((int64_t)a.N * t / n_threads), n1 = (int)((int64_t)a.N * (t + 1) / n_threads);
    th.emplace_back([&a, n0, n1]() { render_range<T>(a, n0, n1); });
  }
  for (auto& t : th) t.join();
}

}  // namespace

// =====================================================================================
// C ABI (ctypes).  _f32 / _f64 twins of everything.
// =====================================================================================
#define ORC_API(T, SFX)                                                                                               \
  extern "C" void orc_project_verts_##SFX(const T* verts, const T* R, const T* Tr, T s, int N, int V, T* out) {       \
    project_verts<T>(verts, R, Tr, s, N, V, out);                                                                     \
  }                                                                                                                   \
  extern "C" void orc_project_verts_bwd_##SFX(const T* verts, const T* R, const T* Tr, T s, int N, int V,             \
                                              const T* g_ndc, T* g_verts, T* g_R, T* g_T) {                           \
    project_verts_bwd<T>(verts, R, Tr, s, N, V, g_ndc, g_verts, g_R, g_T);                                            \
  }                                                                                                                   \
  extern "C" void orc_rasterize_naive_##SFX(const T* fv, const int64_t* first, const int64_t* nf, int N, int H,       \
                                            int W, T blur, int K, int clip, int cull, int64_t* p2f, T* zb, T* bar,    \
                                            T* ds) {                                                                  \
    rasterize_naive<T>(fv, first, nf, N, H, W, blur, K, clip != 0, cull != 0, 0, N, p2f, zb, bar, ds);                \
  }                                                                                                                   \
  extern "C" void orc_rasterize_backward_##SFX(const T* fv, const int64_t* p2f, const T* gz, const T* gb,             \
                                               const T* gd, int N, int H, int W, int K, int clip, T* gfv) {           \
    rasterize_backward<T>(fv, p2f, gz, gb, gd, N, H, W, K, clip != 0, 0, N, gfv);                                     \
  }                                                                                                                   \
  extern "C" void orc_tile_faces_##SFX(const T* fv, const int64_t* first, const int64_t* nf, int N, int H, int W,     \
                                       T blur, int ts, int cull, int cap, int32_t* counts, int32_t* lists) {          \
    tile_faces<T>(fv, first, nf, N, H, W, blur, ts, cull != 0, cap, counts, lists);                                   \
  }                                                                                                                   \
  extern "C" void orc_sample_textures_fwd_##SFX(const T* maps, const T* fuv, const int64_t* p2f, const T* bar,        \
                                                int N, int H, int W, int K, int Ht, int Wt, int C, T* tex) {          \
    sample_textures_fwd<T>(maps, fuv, p2f, bar, N, H, W, K, Ht, Wt, C, tex);                                          \
  }                                                                                                                   \
  extern "C" void orc_sample_textures_bwd_##SFX(const T* maps, const T* fuv, const int64_t* p2f, const T* bar,        \
                                                const T* gtex, int N, int H, int W, int K, int Ht, int Wt, int C,     \
                                                T* gmaps, T* gbar) {                                                  \
    sample_textures_bwd<T>(maps, fuv, p2f, bar, gtex, N, H, W, K, Ht, Wt, C, gmaps, gbar);                            \
  }                                                                                                                   \
  extern "C" void orc_layered_blend_fwd_##SFX(const T* col, const int64_t* p2f, const T* ds, int N, int H, int W,     \
                                              int K, T sigma, const T* bg, int mode, T* image) {                      \
    layered_blend_fwd<T>(col, p2f, ds, N, H, W, K, sigma, bg, mode, image);                                           \
  }                                                                                                                   \
  extern "C" void orc_layered_blend_bwd_##SFX(const T* col, const int64_t* p2f, const T* ds, const T* gimg, int N,    \
                                              int H, int W, int K, T sigma, const T* bg, int mode, T* gcol,           \
                                              T* gds) {                                                               \
    layered_blend_bwd<T>(col, p2f, ds, gimg, N, H, W, K, sigma, bg, mode, gcol, gds);                                 \
  }                                                                                                                   \
  extern "C" void orc_softmax_blend_fwd_##SFX(const T* col, const int64_t* p2f, const T* ds, const T* zb, int N,      \
                                              int H, int W, int K, T sigma, T gamma, const T* bg, T znear, T zfar,    \
                                              T* image) {                                                             \
    softmax_blend_fwd<T>(col, p2f, ds, zb, N, H, W, K, sigma, gamma, bg, znear, zfar, image);                         \
  }                                                                                                                   \
  extern "C" void orc_mesh_regularizers_##SFX(const T* verts, int N, int V, const int32_t* quads, int E,              \
                                              const int32_t* off, const int32_t* idx, T* nsum, T* lsum, T* gN,        \
                                              T* gL) {                                                                \
    mesh_regularizers<T>(verts, N, V, quads, E, off, idx, nsum, lsum, gN, gL);                                        \
  }                                                                                                                   \
  extern "C" void orc_softmax_blend_bwd_##SFX(const T* col, const int64_t* p2f, const T* ds, const T* zb,             \
                                              const T* gimg, int N, int H, int W, int K, T sigma, T gamma,            \
                                              const T* bg, T znear, T zfar, T* gcol, T* gds, T* gzb) {                \
    softmax_blend_bwd<T>(col, p2f, ds, zb, gimg, N, H, W, K, sigma, gamma, bg, znear, zfar, gcol, gds, gzb);          \
  }                                                                                                                   \
  extern "C" void orc_render_fwd_bwd_##SFX(const T* verts, const int64_t* faces, const T* R, const T* Tr,             \
                                           const T* maps, const T* verts_uvs, int N, int V, int F, int H, int W,      \
                                           int K, int Ht, int Wt, T s, T sigma, T blur, const T* bg, int mode,        \
                                           int clip, const T* grad_image, T* image, T* g_verts, T* g_R, T* g_T,       \
                                           T* g_maps, int n_threads) {                                                \
    RenderArgs<T> a{verts, faces, R,     Tr,   maps, verts_uvs, N,         V,          F,     H,       W,   K,        \
                    Ht,    Wt,    s,     sigma, blur, bg,       mode,      clip != 0,  grad_image, image, g_verts,    \
                    g_R,   g_T,   g_maps};                                                                            \
    render_fwd_bwd<T>(a, n_threads);                                                                                  \
  }

ORC_API(float, f32)
ORC_API(double, f64)

extern "C" int orc_hardware_threads() { return (int)std::thread::hardware_concurrency(); }
