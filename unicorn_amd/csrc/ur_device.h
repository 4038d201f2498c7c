// Device-side geometry / shading primitives shared by the kernels of ur_render.hip.
// gfx950 only.  Compiled with -ffp-contract=off: every fp32 operation below is meant to round exactly once, in the
// order written, so that visibility decisions (pix_to_face, tile lists) are reproducible bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ur {

constexpr float kEps = 1e-8f;
constexpr int kTile = 8;  // a wavefront rasterizes one 8x8 pixel tile: lane = pixel

// The hardware's minimum / maximum (one instruction; a compare + select is three issue slots with its hazard wait, and
// the pair evaluation has a dozen of them).  They differ from std::min / std::max -- what the CPU checker uses -- only in
// which zero comes out of (-0, +0) and in dropping a NaN operand; no sign of a zero reaches a result here (no
// division by, square root or copysign of these values: a clamped -0 multiplies, adds and compares like +0), so every
// output still compares equal to the checker's, value for value.
__device__ __forceinline__ float min_s(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float max_s(float a, float b) { return __builtin_fmaxf(a, b); }

// NDC coordinate of the centre of pixel i of an axis with S1 pixels (S2 = the other axis).  +X left / +Y up: the
// caller passes i = S1-1-column.
__device__ __forceinline__ float pix_to_ndc(int i, int S1, int S2) {
  float range = 2.0f;
  if (S1 > S2) range = ((float)S1 / (float)S2) * range;
  const float offset = range / 2.0f;
  return -offset + (range * (float)i + offset) / (float)S1;
}

// The same map with the axis constants taken out (one per wave instead of one per call) and the division by S1 as an
// exact quotient from RN(1/S1) (div_by below): same bits as pix_to_ndc, 6 instructions instead of one or two IEEE
// divisions.  The numerator range * i + offset is positive, so not even the sign of a zero can differ.
struct NdcAxis { float range, offset, s1, is1; };
__device__ __forceinline__ float div_by(float a, float b, float y);
__device__ __forceinline__ NdcAxis ndc_axis(int S1, int S2) {
  NdcAxis a;
  a.range = 2.0f;
  if (S1 > S2) a.range = ((float)S1 / (float)S2) * a.range;
  a.offset = a.range / 2.0f;
  a.s1 = (float)S1;
  a.is1 = 1.0f / a.s1;
  return a;
}
__device__ __forceinline__ float pix_to_ndc(int i, const NdcAxis& a) {
  return -a.offset + div_by(a.range * (float)i + a.offset, a.s1, a.is1);
}

// Correctly rounded a / b from y = RN(1 / b) (computed once per face with an IEEE division): two Newton steps on the
// quotient with exact fma residuals.  q0 = RN(a y) is within 2 ulp; the first step makes it faithful, and for a
// faithful q with y = RN(1/b) Markstein's theorem gives RN(a/b) after the second.  Five instructions instead of the
// ~10 of the full IEEE sequence; the only observable difference is the sign of a zero quotient.  Checked against
// `/` bit for bit by ur_selftest_division and, on whole scenes, by every bit-exact parity test.
__device__ __forceinline__ float div_by(float a, float b, float y) {
  float q = a * y;
  float r = __builtin_fmaf(-b, q, a);
  q = __builtin_fmaf(r, y, q);
  r = __builtin_fmaf(-b, q, a);
  return __builtin_fmaf(r, y, q);
}

// Per-face record prepared once per call (k_face_prep): everything that does not depend on the pixel.
//   rec[5*f+0] = v0x v0y v1x v1y
//   rec[5*f+1] = v2x v2y z0  z1
//   rec[5*f+2] = z2  denom 1/denom l2(v0v1)           denom = edge(v2; v0, v1) + 1e-8 ; reciprocals are RN(1/x)
//   rec[5*f+3] = l2(v0v2) l2(v1v2) 1/l2(v0v1) 1/l2(v0v2)
//   rec[5*f+4] = 1/l2(v1v2) d12x d12y zmin            d12 = v2 - v1 ; zmin = nearest corner depth
//   bbox[f]    = xlo xhi ylo yhi                      blur-expanded; culled faces get xlo = +inf, xhi = -inf
struct Face {
  float v0x, v0y, v1x, v1y, v2x, v2y, z0, z1, z2, denom, idenom, l01, l02, l12, il01, il02, il12, d12x, d12y, zmin;
};

__device__ __forceinline__ Face unpack_face(const float4 a, const float4 b, const float4 c, const float4 d,
                                            const float4 e) {
  Face f;
  f.v0x = a.x; f.v0y = a.y; f.v1x = a.z; f.v1y = a.w;
  f.v2x = b.x; f.v2y = b.y; f.z0 = b.z; f.z1 = b.w;
  f.z2 = c.x; f.denom = c.y; f.idenom = c.z; f.l01 = c.w;
  f.l02 = d.x; f.l12 = d.y; f.il01 = d.z; f.il02 = d.w;
  f.il12 = e.x; f.d12x = e.y; f.d12y = e.z; f.zmin = e.w;
  return f;
}

__device__ __forceinline__ Face load_face(const float4* __restrict__ rec, int fid) {
  return unpack_face(rec[5 * (size_t)fid + 0], rec[5 * (size_t)fid + 1], rec[5 * (size_t)fid + 2],
                     rec[5 * (size_t)fid + 3], rec[5 * (size_t)fid + 4]);
}

// squared distance from p to segment a-b ; (bax, bay) = b - a, l2 = |b - a|^2 and il2 = RN(1/l2)
__device__ __forceinline__ float seg_dist(float px, float py, float ax, float ay, float bx, float by, float bax,
                                          float bay, float l2, float il2) {
  if (l2 <= kEps) {
    const float dx = px - bx, dy = py - by;
    return dx * dx + dy * dy;
  }
  const float pax = px - ax, pay = py - ay;
  const float t = div_by(bax * pax + bay * pay, l2, il2);
  const float tt = min_s(max_s(t, 0.0f), 1.0f);
  const float qx = ax + tt * bax, qy = ay + tt * bay;
  const float dx = px - qx, dy = py - qy;
  return dx * dx + dy * dy;
}

struct Eval {
  float pz, sd, c0, c1, c2;  // depth, signed squared distance, output barycentrics
  float w0, w1, w2;          // unclipped barycentrics
  float d01, d02, d12;       // the three edge distances
  bool inside;
};

// The (pixel, face) test.  Returns false when the face does not contribute to the pixel.  The bbox test against the
// blur-expanded box must have been done by the caller (it needs the bbox, not the record).
__device__ __forceinline__ bool eval_face(const Face& f, float px, float py, float blur, bool clip, Eval& o) {
  const float d01x = f.v1x - f.v0x, d01y = f.v1y - f.v0y, d20x = f.v0x - f.v2x, d20y = f.v0y - f.v2y;
  const float e0 = (px - f.v1x) * f.d12y - (py - f.v1y) * f.d12x;  // edge(p; v1, v2)
  const float e1 = (px - f.v2x) * d20y - (py - f.v2y) * d20x;      // edge(p; v2, v0)
  const float e2 = (px - f.v0x) * d01y - (py - f.v0y) * d01x;      // edge(p; v0, v1)
  o.w0 = div_by(e0, f.denom, f.idenom); o.w1 = div_by(e1, f.denom, f.idenom); o.w2 = div_by(e2, f.denom, f.idenom);
  float c0 = o.w0, c1 = o.w1, c2 = o.w2;
  if (clip) {
    c0 = max_s(c0, 0.0f); c1 = max_s(c1, 0.0f); c2 = max_s(c2, 0.0f);
    const float s = max_s(c0 + c1 + c2, 1e-5f);
    const float ys = 1.0f / s;  // one IEEE division, then three exact quotients (div_by): same bits as c / s
    c0 = div_by(c0, s, ys); c1 = div_by(c1, s, ys); c2 = div_by(c2, s, ys);
  }
  float pz = c0 * f.z0 + c1 * f.z1 + c2 * f.z2;
  const bool front = !(pz < 0.0f);  // (no early exit: a face behind the camera is rare, a branch per pair is not free)
  pz = pz + 0.0f;  // -0 -> +0: depth keys are compared as unsigned bit patterns
  o.d01 = seg_dist(px, py, f.v0x, f.v0y, f.v1x, f.v1y, d01x, d01y, f.l01, f.il01);
  o.d02 = seg_dist(px, py, f.v0x, f.v0y, f.v2x, f.v2y, -d20x, -d20y, f.l02, f.il02);
  o.d12 = seg_dist(px, py, f.v1x, f.v1y, f.v2x, f.v2y, f.d12x, f.d12y, f.l12, f.il12);
  const float dist = min_s(min_s(o.d01, o.d02), o.d12);
  o.inside = (o.w0 > 0.0f) && (o.w1 > 0.0f) && (o.w2 > 0.0f);
  o.pz = pz; o.sd = o.inside ? -dist : dist;
  o.c0 = c0; o.c1 = c1; o.c2 = c2;
  return front && (o.inside || !(dist >= blur));
}

// ---- texture sampling: grid_sample(flip_y(map), 2uv-1, bilinear, align_corners=True, padding 'border') ----------
struct Bilin {
  int x0, y0;               // top-left texel in the FLIPPED map; x1 = x0+1, y1 = y0+1 may fall outside (weight 0)
  float wnw, wne, wsw, wse;
  float ix, iy;             // clipped source coordinates
  float mx, my;             // d(ix)/d(gx), d(iy)/d(gy) including the border-clip mask
};

__device__ __forceinline__ Bilin bilin_setup(float u, float v, int Ht, int Wt) {
  Bilin b;
  const float gx = u * 2.0f - 1.0f, gy = v * 2.0f - 1.0f;
  float ix = ((gx + 1.0f) / 2.0f) * (float)(Wt - 1);
  float iy = ((gy + 1.0f) / 2.0f) * (float)(Ht - 1);
  b.mx = (float)(Wt - 1) / 2.0f;
  b.my = (float)(Ht - 1) / 2.0f;
  if (ix <= 0.0f) { ix = 0.0f; b.mx = 0.0f; } else if (ix >= (float)(Wt - 1)) { ix = (float)(Wt - 1); b.mx = 0.0f; }
  if (iy <= 0.0f) { iy = 0.0f; b.my = 0.0f; } else if (iy >= (float)(Ht - 1)) { iy = (float)(Ht - 1); b.my = 0.0f; }
  b.ix = ix; b.iy = iy;
  const float fx = floorf(ix), fy = floorf(iy);
  b.x0 = (int)fx; b.y0 = (int)fy;
  const float x1 = (float)(b.x0 + 1), y1 = (float)(b.y0 + 1);
  b.wnw = (x1 - ix) * (y1 - iy);
  b.wne = (ix - fx) * (y1 - iy);
  b.wsw = (x1 - ix) * (iy - fy);
  b.wse = (ix - fx) * (iy - fy);
  return b;
}

// flat offsets (in floats) of the four texels inside one (Ht,Wt,3) map, or -1 when outside
__device__ __forceinline__ void bilin_offsets(const Bilin& b, int Ht, int Wt, int& onw, int& one, int& osw, int& ose) {
  const bool x1in = (b.x0 + 1) < Wt, y1in = (b.y0 + 1) < Ht;  // x0, y0 are always inside after the clip
  const int r0 = (Ht - 1 - b.y0) * Wt, r1 = (Ht - 2 - b.y0) * Wt;  // flipped rows
  onw = (r0 + b.x0) * 3;
  one = x1in ? (r0 + b.x0 + 1) * 3 : -1;
  osw = y1in ? (r1 + b.x0) * 3 : -1;
  ose = (x1in && y1in) ? (r1 + b.x0 + 1) * 3 : -1;
}

__device__ __forceinline__ void sample_rgb(const float* __restrict__ map, const Bilin& b, int Ht, int Wt, float& r,
                                           float& g, float& bl) {
  int onw, one, osw, ose;
  bilin_offsets(b, Ht, Wt, onw, one, osw, ose);
  // all twelve loads are issued unconditionally (a texel outside the map reads the north-west one instead and is
  // replaced by 0 afterwards): straight-line code, the loads of several samples can be in flight together
  const int ene = one >= 0 ? one : onw, esw = osw >= 0 ? osw : onw, ese = ose >= 0 ? ose : onw;
  float t[4][3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    t[0][c] = map[onw + c]; t[1][c] = map[ene + c]; t[2][c] = map[esw + c]; t[3][c] = map[ese + c];
  }
  float acc[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float a = 0.0f;
    a += t[0][c] * b.wnw;
    a += (one >= 0 ? t[1][c] : 0.0f) * b.wne;
    a += (osw >= 0 ? t[2][c] : 0.0f) * b.wsw;
    a += (ose >= 0 ? t[3][c] : 0.0f) * b.wse;
    acc[c] = a;
  }
  r = acc[0]; g = acc[1]; bl = acc[2];
}

// alpha of layered_rgb_blend (renderer.py:186-191); `valid` = pix_to_face >= 0
__device__ __forceinline__ float layer_alpha(float d, bool valid, float sigma, int mode) {
  if (!valid) return 0.0f;
  if (mode == 0) return d <= 0.0f ? 1.0f : 0.0f;
  if (mode == 1) return expf(-max_s(d, 0.0f) / sigma);
  return 1.0f / (1.0f + expf(d / sigma));
}

// d(alpha)/d(dist) given alpha
__device__ __forceinline__ float layer_dalpha(float d, float a, bool valid, float sigma, int mode) {
  if (!valid || mode == 0) return 0.0f;
  if (mode == 1) return (d >= 0.0f) ? a * (-1.0f / sigma) : 0.0f;
  return a * (1.0f - a) * (-1.0f / sigma);
}

// true only if layer_alpha() == 1 exactly and layer_dalpha() == 0 for this signed distance (a sufficient test that
// needs no exp): such a layer hides every layer behind it from the image and from all gradients.
__device__ __forceinline__ bool hides_rest(float d, float sigma, int mode) {
  if (mode == 0) return d <= 0.0f;
  if (mode == 1) return d < 0.0f;           // exp(-0/sigma) == 1, clamp gradient 0
  return d < -20.0f * sigma;                // exp(d/sigma) < 2^-25: 1/(1+e) rounds to 1, a(1-a) == 0
}

}  // namespace ur
