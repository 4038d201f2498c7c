// ur_render.hip -- hand-written gfx950 (MI355X) kernels + the C ABI of include/unicorn_render.h.
//
// Execution model (DESIGN.md 5 has the long version):
//   * k_face_prep: one thread per face -> bounding box, per-face record (corners, denominators, RN reciprocals), and
//     a per-bin load count (one-call path, equal-sized meshes: k_project_prep_mesh does this together with the camera
//     projection and leaves per-bin face masks); k_bin_order sorts the 16x16-pixel bins of all meshes heaviest first;
//   * k_raster<MODE, BIGK, MASKS, G8>: one workgroup per bin, taken from that schedule: four waves of 64 pixels, or
//     (G8, launches too small to fill the chip) eight waves of 32.  Coarse step: the workgroup scans the mesh's face
//     boxes (up to 8 coalesced float4 per lane) or reads the bin's face mask, ballot + popcount prefix packs survivors
//     into one bin list in LDS -- no atomics, deterministic; rounds of 2048 faces, the list cannot overflow.  Fine step: 64 bin
//     faces at a time staged in LDS (double buffered).  MODE 0 (fragments): lane = pixel, faces walked one by one.
//     MODE 2 (fused shading): (face, pixel) pairs dealt over the lanes, hits appended to per-pixel lists with LDS
//     integer atomics.  MODE 1: tile lists exported;
//   * per-pixel top-K: (depth, face) keys packed as u64 in LDS, [slot][pixel]; z >= 0, so unsigned order == (z, face)
//     order; lists past the LDS cap continue in a global spill area;
//   * MODE 2 tail: sort, (pixel, layer) items dealt over the lanes, UV sample + layered blend, layer records saved
//     for k_render_bwd, which deals the same items again and scatters gradients through f64 LDS tables.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <type_traits>
#include <vector>

#include "../../include/unicorn_render.h"
#include "ur_device.h"

using namespace ur;

extern "C" size_t ur_workspace_bytes(int64_t total_faces, int N, int H, int W, int K);

// Device memory of the allocation-free entry point (ur_rasterize_meshes_static): one copy per GPU, held from module
// load to process exit.  In host code this name is only a handle for hipGetSymbolAddress -- kernels and memsets are
// given the per-device address kept in g_scratch_ptr, never the name.
static __device__ __align__(256) unsigned char scratch[UR_STATIC_SCRATCH_BYTES];

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// ---- opt-in kernel timing (ur_profile_begin / ur_profile_end): HIP events recorded on the launch stream right around
// the two dominant kernels.  Process-wide (the autograd engine runs the backward on another thread than the forward).
struct ProfState {
  std::vector<hipEvent_t> ev;   // 2 per slot: start, stop
  std::vector<int> kind;        // UR_PROFILE_RASTER_FWD / UR_PROFILE_RENDER_BWD
  std::atomic<int> next{0};
  std::atomic<bool> on{false};
};
ProfState g_prof;

struct ProfScope {  // records the slot's events around a kernel launch when profiling is on
  hipStream_t st;
  int slot = -1;
  ProfScope(int kind, hipStream_t s) : st(s) {
    if (!g_prof.on.load(std::memory_order_acquire)) return;
    const int i = g_prof.next.fetch_add(1);
    if (i >= (int)g_prof.kind.size()) return;
    slot = i;
    g_prof.kind[i] = kind;
    (void)hipEventRecord(g_prof.ev[2 * i], st);
  }
  ~ProfScope() {
    if (slot >= 0) (void)hipEventRecord(g_prof.ev[2 * slot + 1], st);
  }
};

// per-device address of `scratch`, looked up once per device with that device current
std::mutex g_scratch_mu;
void* g_scratch_ptr[64] = {};

int static_scratch(void** out) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fail(UR_ERR_LAUNCH, "no current device");
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !p) return fail(UR_ERR_LAUNCH, "hipGetSymbolAddress(scratch): %s", hipGetErrorString(e));
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return UR_OK;
}

constexpr int kMaxDevices = 64;

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(UR_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return UR_OK;
}

// Kernels that want more than 64 KB of dynamic LDS must be told so once per (kernel, device).  `slot` is a small
// table per call site: the largest size configured so far on each device (process-wide, relaxed atomics: a race only
// repeats the idempotent call).
struct LdsSlot { std::atomic<size_t> done[kMaxDevices]; };

int ensure_dynamic_lds(const void* kernel, size_t bytes, LdsSlot& slot) {
  if (bytes <= 64 * 1024) return UR_OK;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = -1;
  if (dev >= 0 && slot.done[dev].load(std::memory_order_relaxed) >= bytes) return UR_OK;
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return fail(UR_ERR_LAUNCH, "hipFuncSetAttribute(%zu B LDS): %s", bytes, hipGetErrorString(e));
  if (dev >= 0) slot.done[dev].store(bytes, std::memory_order_relaxed);
  return UR_OK;
}

// Lanes of one wave hand values to each other through LDS (one lane writes a slot, another reads it) without a
// barrier: the hardware executes a wave's LDS operations in order.  The COMPILER must be told as well -- to the C++
// memory model an unsynchronised write by another lane is a data race it may ignore, and it does: it forwards a lane's
// own earlier store to its later load of the same slot and drops the load (seen on a marks[] read, which then missed
// the other lanes' marks).  A release + acquire fence at wavefront scope between the writes and the reads says so and
// emits no instruction.
__device__ __forceinline__ void wave_lds_handoff() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Wave-wide inclusive scans on the VALU's data-parallel-primitive lanes (row shifts inside a 16-lane row, then the
// two row broadcasts): six VALU operations, where a __shfl_up ladder is six dependent trips through the LDS crossbar.
// `ident` is the operation's neutral element (what a lane combines with when it has no source lane).
#define UR_SCAN_STEP(CTRL, ROWMASK) v = op(v, __builtin_amdgcn_update_dpp(ident, v, CTRL, ROWMASK, 0xf, false))
template <class Op> __device__ __forceinline__ int wave_scan(int v, int ident, Op op) {
  UR_SCAN_STEP(0x111, 0xf);  // row_shr:1
  UR_SCAN_STEP(0x112, 0xf);  // row_shr:2
  UR_SCAN_STEP(0x114, 0xf);  // row_shr:4
  UR_SCAN_STEP(0x118, 0xf);  // row_shr:8
  UR_SCAN_STEP(0x142, 0xa);  // row_bcast:15 -> rows 1, 3
  UR_SCAN_STEP(0x143, 0xc);  // row_bcast:31 -> rows 2, 3
  return v;
}
#undef UR_SCAN_STEP
__device__ __forceinline__ int wave_scan_add(int v) {
  return wave_scan(v, 0, [](int a, int b) { return a + b; });
}
__device__ __forceinline__ int wave_scan_max(int v) {  // v >= 0
  return wave_scan(v, 0, [](int a, int b) { return max(a, b); });
}
// maximum over the wave of non-negative ints, in every lane
__device__ __forceinline__ int wave_max(int v) { return __builtin_amdgcn_readlane(wave_scan_max(v), 63); }
// sum over the wave (valid in lane 63; summation order fixed by the scan network: deterministic)
__device__ __forceinline__ float wave_sum(float v) {
  return __int_as_float(wave_scan(__float_as_int(v), 0, [](int a, int b) {
    return __float_as_int(__int_as_float(a) + __int_as_float(b)); }));
}
__device__ __forceinline__ float wave_fmin(float v) {
  const int r = wave_scan(__float_as_int(v), __float_as_int(__builtin_inff()),
                          [](int a, int b) { return __float_as_int(fminf(__int_as_float(a), __int_as_float(b))); });
  return __int_as_float(__builtin_amdgcn_readlane(r, 63));
}
__device__ __forceinline__ float wave_fmax(float v) {
  const int r = wave_scan(__float_as_int(v), __float_as_int(-__builtin_inff()),
                          [](int a, int b) { return __float_as_int(fmaxf(__int_as_float(a), __int_as_float(b))); });
  return __int_as_float(__builtin_amdgcn_readlane(r, 63));
}

// Value of lane (lane ^ D), D a power of two: the exchange step of the cross-lane sorting networks.  Distances up to 8
// stay inside a 16-lane row and are one or two data-parallel-primitive moves on the VALU (quad permutes for 1 and 2,
// two bank-masked row shifts for 4, a row rotation by 8 -- 16 lanes rotated by 8 IS the xor); 16 and 32 are gfx950's
// row-pair swaps followed by a select.  A __shfl_xor is a round trip through the LDS crossbar per 32 bits instead, and
// a bitonic network is a chain of them: the sorts of a silhouette bin's long lists spent their time waiting on those.
template <int D> __device__ __forceinline__ unsigned lane_xor(unsigned v) {
  static_assert(D == 1 || D == 2 || D == 4 || D == 8 || D == 16 || D == 32, "lane_xor: D");
  if constexpr (D == 1) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
  } else if constexpr (D == 2) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);  // quad_perm [2,3,0,1]
  } else if constexpr (D == 4) {
    int r = __builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xf, 0x5, false);  // row_shl:4 -> lanes 0-3, 8-11 of a row
    r = __builtin_amdgcn_update_dpp(r, (int)v, 0x114, 0xf, 0xa, false);      // row_shr:4 -> lanes 4-7, 12-15
    return (unsigned)r;
  } else if constexpr (D == 8) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, true);  // row_ror:8
  } else if constexpr (D == 16) {
    const auto p = __builtin_amdgcn_permlane16_swap(v, v, false, false);  // p[0]: rows 0,0,2,2 ; p[1]: rows 1,1,3,3
    return (__lane_id() & 16) ? p[0] : p[1];
  } else {
    const auto p = __builtin_amdgcn_permlane32_swap(v, v, false, false);  // p[0]: lower half twice ; p[1]: upper half twice
    return (__lane_id() & 32) ? p[0] : p[1];
  }
}
template <int D> __device__ __forceinline__ unsigned long long lane_xor(unsigned long long v) {
  return ((unsigned long long)lane_xor<D>((unsigned)(v >> 32)) << 32) | lane_xor<D>((unsigned)v);
}
// the same with the distance as a value a fully unrolled loop has made constant
__device__ __forceinline__ unsigned long long lane_xor_d(unsigned long long v, int d) {
  switch (d) {
    case 1: return lane_xor<1>(v);
    case 2: return lane_xor<2>(v);
    case 4: return lane_xor<4>(v);
    case 8: return lane_xor<8>(v);
    case 16: return lane_xor<16>(v);
    default: return lane_xor<32>(v);
  }
}

constexpr int kPrepBins = 4096;  // bins per image the face-prep kernel counts in LDS (16 KB); beyond: global atomics
// Face masks (k_project_prep_mesh -> k_raster<2, ., true>): per (mesh, bin, part) 256 bits, bit j = face part * 256 + j
// of the mesh passes the bin's box test.  Up to kMaskBins bins per image, at most kMaskAxis of them along either axis.
constexpr int kMaskBins = 1024;
constexpr int kMaskAxis = 256;
constexpr int kEmptyBin = 0x40000000;  // flag on a schedule entry: no face box reaches the bin
constexpr int kBinIndex = 0x3fffffff;

__device__ __forceinline__ bool bbox_overlap(const float4 bb, float xc_lo, float xc_hi, float yc_lo, float yc_hi) {
  // bb = xlo xhi ylo yhi ; reject iff the whole centre interval is on one side of the box
  return !(xc_hi < bb.x || xc_lo > bb.y || yc_hi < bb.z || yc_lo > bb.w);
}

// ------------------------------------------------------------------------------------------------------------------
// face records
// ------------------------------------------------------------------------------------------------------------------
// One face: blur-expanded box, record, nearest depth (layout in ur_device.h).  Returns true when the face is culled.
__device__ __forceinline__ bool face_record(const float c[9], float r, int cull, float4& bb, float4 rc[5]) {
  const float v0x = c[0], v0y = c[1], z0 = c[2], v1x = c[3], v1y = c[4], z1 = c[5], v2x = c[6], v2y = c[7], z2 = c[8];
  const float zmax = max_s(max_s(z0, z1), z2);
  const float face_area = (v0x - v1x) * (v2y - v1y) - (v0y - v1y) * (v2x - v1x);  // edge(v0; v1, v2)
  // (zmax < eps: SPEC DECISION, a face wholly behind the camera plane; the min-z alternative is on record in DESIGN.md 2
  // and beside the same test of the CPU checker -- change both together)
  const bool culled = (zmax < kEps) || (face_area <= kEps && face_area >= -kEps) || (cull && face_area < 0.0f);
  float xlo = min_s(min_s(v0x, v1x), v2x) - r, xhi = max_s(max_s(v0x, v1x), v2x) + r;
  const float ylo = min_s(min_s(v0y, v1y), v2y) - r, yhi = max_s(max_s(v0y, v1y), v2y) + r;
  if (culled) { xlo = __builtin_inff(); xhi = -__builtin_inff(); }
  bb = make_float4(xlo, xhi, ylo, yhi);
  const float denom = ((v2x - v0x) * (v1y - v0y) - (v2y - v0y) * (v1x - v0x)) + kEps;  // edge(v2; v0, v1) + eps
  const float d01x = v1x - v0x, d01y = v1y - v0y, d12x = v2x - v1x, d12y = v2y - v1y;
  const float d02x = v2x - v0x, d02y = v2y - v0y;
  const float l01 = d01x * d01x + d01y * d01y, l02 = d02x * d02x + d02y * d02y, l12 = d12x * d12x + d12y * d12y;
  rc[0] = make_float4(v0x, v0y, v1x, v1y);
  rc[1] = make_float4(v2x, v2y, z0, z1);
  rc[2] = make_float4(z2, denom, 1.0f / denom, l01);
  rc[3] = make_float4(l02, l12, 1.0f / l01, 1.0f / l02);
  rc[4] = make_float4(1.0f / l12, d12x, d12y, min_s(min_s(z0, z1), z2));
  return culled;
}

// pixel rectangle (columns c0..c1, rows r0..r1, a pixel of slack either way) a blur-expanded box may touch.  It DECIDES
// geometry: a bin outside it gets neither a load count (a bin of load 0 is flagged kEmptyBin and skipped by both big
// kernels) nor, on the mask path, the face's mask bit.  It must therefore be a superset of bbox_overlap()'s verdicts for
// ANY box, including the huge or non-finite ones a vertex near the camera plane produces: the pixel coordinates are
// clamped as floats before the cast (an unclamped cast saturates and the +-1 then wraps), and a NaN bound -- which
// fails every comparison of bbox_overlap, i.e. passes every bin -- opens its side of the rectangle completely.
__device__ __forceinline__ void box_pixels(const float4 bb, int H, int W, int& c0, int& c1, int& r0, int& r1) {
  const float rx = (W > H) ? 2.0f * (float)W / (float)H : 2.0f, ry = (H > W) ? 2.0f * (float)H / (float)W : 2.0f;
  // column c has x = rx/2 - rx (c + 0.5) / W  (+X left), row likewise
  // lower ends: NaN -> -1 (fmaxf drops the NaN) -> 0 ; upper ends: NaN -> S (fminf drops it) -> S - 1
  const float fc0 = fminf(fmaxf((0.5f * rx - bb.y) * (float)W / rx - 0.5f, -1.0f), (float)W);
  const float fc1 = fmaxf(fminf((0.5f * rx - bb.x) * (float)W / rx - 0.5f, (float)W), -2.0f);
  const float fr0 = fminf(fmaxf((0.5f * ry - bb.w) * (float)H / ry - 0.5f, -1.0f), (float)H);
  const float fr1 = fmaxf(fminf((0.5f * ry - bb.z) * (float)H / ry - 0.5f, (float)H), -2.0f);
  c0 = max(0, (int)floorf(fc0) - 1);
  c1 = min(W - 1, (int)ceilf(fc1) + 1);
  r0 = max(0, (int)floorf(fr0) - 1);
  r1 = min(H - 1, (int)ceilf(fr1) + 1);
}

__global__ void __launch_bounds__(256) k_face_prep(const float* __restrict__ fv, int64_t NF, float r, int cull,
                                                   float4* __restrict__ bbox, float4* __restrict__ rec,
                                                   float* __restrict__ fz, const int64_t* __restrict__ first, int N,
                                                   int H, int W, int* __restrict__ binload) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < NF;
  bool culled = true;
  float4 bb = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (live) {
    const float* p = fv + i * 9;
    const float c[9] = {p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8]};
    float4 rc[5];
    culled = face_record(c, r, cull, bb, rc);
    bbox[i] = bb;
    fz[i] = rc[4].w;
#pragma unroll
    for (int q = 0; q < 5; ++q) rec[5 * i + q] = rc[q];
  }
  // ---- load of the workgroup schedule: +1 on every 16x16 bin the expanded box may touch (box_pixels: a superset of
  //      the bin test; a bin of load 0 is skipped by the big kernels).  The workgroup first counts in LDS for the
  //      mesh of its first face (consecutive faces nearly always share the mesh), then flushes the non-zero bins.
  if (!binload) return;
  const int nbx = (W + 15) >> 4, nby = (H + 15) >> 4, nbins = nbx * nby;
  __shared__ int s_load[kPrepBins];
  __shared__ int s_mesh;
  const bool use_lds = nbins <= kPrepBins;
  if (use_lds) {
    for (int j = threadIdx.x; j < nbins; j += 256) s_load[j] = 0;
    if (threadIdx.x == 0) {
      const int64_t i0 = (int64_t)blockIdx.x * blockDim.x;
      int lo = 0, hi = N - 1;  // mesh of face i0: last n with first[n] <= i0
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (first[mid] <= i0) lo = mid; else hi = mid - 1;
      }
      s_mesh = lo;
    }
    __syncthreads();
  }
  if (live && !culled) {
    int lo = 0, hi = N - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (first[mid] <= i) lo = mid; else hi = mid - 1;
    }
    int c0, c1, r0, r1;
    box_pixels(bb, H, W, c0, c1, r0, r1);
    const bool local = use_lds && lo == s_mesh;
    for (int by = r0 >> 4; by <= (r1 >> 4); ++by)
      for (int bx = c0 >> 4; bx <= (c1 >> 4); ++bx) {
        if (local) atomicAdd(&s_load[by * nbx + bx], 1);
        else atomicAdd(&binload[(int64_t)lo * nbins + by * nbx + bx], 1);
      }
  }
  if (use_lds) {
    __syncthreads();
    for (int j = threadIdx.x; j < nbins; j += 256) {
      const int v = s_load[j];
      if (v) atomicAdd(&binload[(int64_t)s_mesh * nbins + j], v);
    }
  }
}

// Workgroup schedule: bins sorted by descending load estimate (counting sort over 256 load classes, one workgroup).
// The long-running bins -- silhouettes, where many faces pile up -- then start first and the cheap ones fill the gaps
// at the end, instead of a late heavy bin stretching the kernel.  Any permutation gives the same results.  A bin whose
// load is zero is flagged (kEmptyBin): the load counts every face whose box, taken a pixel wider, reaches the bin, so no
// face box passes such a bin's test and the kernels skip their face scan / row sweep for it.
// Meshes of many faces leave many partial counts per bin (one per 256 faces): summed here by as many workgroups as it
// takes, in place (the total lands in the mesh's part-0 slot), so that the one-workgroup sort below reads one number
// per bin.  20480-face meshes: 80 parts, the sort kernel 23 -> 8 us for 16 meshes, 8x that for 128.
__global__ void __launch_bounds__(256) k_bin_sum(int* __restrict__ binload, int parts, int bins_per_mesh) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= bins_per_mesh) return;
  int* p = binload + (int64_t)blockIdx.y * parts * bins_per_mesh + b;
  int sum = 0;
  for (int q0 = 0; q0 < parts; q0 += 8) {
    int t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = (q0 + u < parts) ? p[(int64_t)(q0 + u) * bins_per_mesh] : 0;
    sum += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
  }
  p[0] = sum;
}
constexpr int kBinSumParts = 32;  // from this many parts per mesh on, k_bin_sum runs before the sort (at 20 parts --
                                  // 5120-face meshes -- the two kernels together take what the sort alone did)

// `parts` partial counts per (mesh, bin), laid out [mesh][part][bin]; parts == 1 with mesh_stride > bins_per_mesh: the
// totals k_bin_sum left in the part-0 slots of that layout
__global__ void __launch_bounds__(1024) k_bin_order(const int* __restrict__ binload, int n_bins, int* __restrict__ order,
                                                    int parts, int bins_per_mesh, int mesh_stride) {
  __shared__ int s_hist[256];
  __shared__ int s_wsum[4];
  __shared__ int s_max;
  const int tid = threadIdx.x;
  if (tid < 256) s_hist[tid] = 0;
  if (tid == 0) s_max = 1;
  // the first four loads of every thread stay in registers for the three passes (covers 4096 bins)
  // load of bin i: binload[i], or -- when the per-mesh prep kernel left `parts` partial counts per mesh, laid out
  // [mesh][part][bin] -- their sum
  auto load_of = [&](int i) {
    if (parts == 1 && mesh_stride == bins_per_mesh) return binload[i];
    const int m = i / bins_per_mesh, b = i - m * bins_per_mesh;
    if (parts == 1) return binload[(int64_t)m * mesh_stride + b];
    const int* p = binload + ((int64_t)m * parts) * bins_per_mesh + b;
    int sum = 0;
    for (int q0 = 0; q0 < parts; q0 += 8) {  // eight independent loads in flight
      int t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = (q0 + u < parts) ? p[(int64_t)(q0 + u) * bins_per_mesh] : 0;
      sum += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    }
    return sum;
  };
  int v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = (tid + q * 1024 < n_bins) ? load_of(tid + q * 1024) : 0;
  auto each = [&](auto fn) {  // fn(load, bin index) over this thread's bins
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (tid + q * 1024 < n_bins) fn(v[q], tid + q * 1024);
    for (int i = tid + 4096; i < n_bins; i += 1024) fn(load_of(i), i);
  };
  __syncthreads();
  int mx = 1;
  each([&](int x, int) { mx = max(mx, x); });
  mx = wave_max(mx);
  if ((tid & 63) == 0) atomicMax(&s_max, mx);
  __syncthreads();
  const float scale = 127.0f / (float)s_max;
  // classes 0..127: loaded bins, heaviest first; classes 128..255: the empty bins, spread by index so that they do
  // not all queue on one LDS counter
  auto cls = [&](int x, int i) { return x <= 0 ? 128 + (i & 127) : 127 - min(127, (int)((float)x * scale)); };
  each([&](int x, int i) { atomicAdd(&s_hist[cls(x, i)], 1); });
  __syncthreads();
  {  // exclusive prefix over the 256 classes: scan inside each of the first four waves, then add the waves before
    const int h = tid < 256 ? s_hist[tid] : 0;
    const int incl = wave_scan_add(h);
    if (tid < 256 && (tid & 63) == 63) s_wsum[tid >> 6] = incl;
    __syncthreads();
    if (tid < 256) {
      int off = 0;
      for (int w = 0; w < (tid >> 6); ++w) off += s_wsum[w];
      s_hist[tid] = off + incl - h;
    }
  }
  __syncthreads();
  each([&](int x, int i) { order[atomicAdd(&s_hist[cls(x, i)], 1)] = x <= 0 ? (i | kEmptyBin) : i; });
}

// camera: view = X R + T, x_ndc = sx x / z, y_ndc = sy y / z, depth = view z (operation order fixed: part of the parity contract)
__device__ __forceinline__ void project_vertex(const float* __restrict__ p, const float* __restrict__ r,
                                               const float* __restrict__ t, float sx, float sy, float out[3]) {
  const float vx = ((p[0] * r[0] + p[1] * r[3]) + p[2] * r[6]) + t[0];
  const float vy = ((p[0] * r[1] + p[1] * r[4]) + p[2] * r[7]) + t[1];
  const float vz = ((p[0] * r[2] + p[1] * r[5]) + p[2] * r[8]) + t[2];
  out[0] = (sx * vx) / vz;
  out[1] = (sy * vy) / vz;
  out[2] = vz;
}

constexpr int kMeshPrepMaxFaces = 32768;  // up to here the per-mesh prep kernel (<= 128 workgroups per mesh)

// workgroups per mesh of k_project_prep_mesh = partial load counts per (mesh, bin); 1 when that kernel is not used
__host__ __device__ inline int prep_parts(int64_t total_faces, int N) {
  const int64_t F = N > 0 ? (total_faces + N - 1) / N : 0;
  return (F > 0 && F <= kMeshPrepMaxFaces) ? (int)((F + 255) / 256) : 1;
}

// Equal-sized meshes (the reference's case: one template topology), ONE launch for everything that comes before the
// schedule: camera projection + face gather (= k_project_faces), face records (= k_face_prep, same arithmetic, same
// bits) and the per-bin load count.  Workgroup (part, mesh) takes 256 consecutive faces of its mesh, counts in LDS and
// STORES its partial counts ([mesh][part][bin]; k_bin_order adds the parts up): no global atomics, nothing to zero
// beforehand.  (A per-mesh sort interleaved rank by rank, which would also save the k_bin_order launch, was tried: it
// costs the two big kernels 5 us each -- strict heaviest-first matters.)  Optionally the workgroups also clear their
// share of the two gradient accumulators the backward will add into (saves the backward a fill launch).
__global__ void __launch_bounds__(256) k_project_prep_mesh(
    const float* __restrict__ verts, const int64_t* __restrict__ faces, int64_t fstride, const float* __restrict__ R,
    const float* __restrict__ T, float sx, float sy, int N, int M, int V, int F, float r, int cull, int H, int W,
    float* __restrict__ face_verts, float4* __restrict__ bbox, float4* __restrict__ rec, float* __restrict__ fz,
    int* __restrict__ binload, unsigned* __restrict__ masks, float* __restrict__ zero_a, int zero_a_n,
    float* __restrict__ zero_b, int zero_b_n) {
  __shared__ int s_load[kPrepBins];
  __shared__ unsigned s_mask[kMaskBins * 8];
  __shared__ float s_ext[4 * kMaskAxis];  // per bin column: x centres lo / hi ; per bin row: y centres lo / hi
  const int part = blockIdx.x, parts = gridDim.x, n = blockIdx.y, tid = threadIdx.x;
  const int m = n % M;  // the mesh this image shows (M == N: one mesh per image; M < N: M meshes tiled over the poses)
  const int nbx = (W + 15) >> 4, nby = (H + 15) >> 4, nbins = nbx * nby;
  for (int j = tid; j < nbins; j += 256) s_load[j] = 0;
  if (masks) {
    // With masks the per-bin test is k_raster's own (the bin's pixel-centre extents against the box), so that the
    // mask IS the bin list and the load its length.
    for (int j = tid; j < nbins * 8; j += 256) s_mask[j] = 0u;
    for (int j = tid; j < nbx; j += 256) {
      s_ext[j] = pix_to_ndc(W - 1 - (min(W, j * 16 + 16) - 1), W, H);
      s_ext[kMaskAxis + j] = pix_to_ndc(W - 1 - j * 16, W, H);
    }
    for (int j = tid; j < nby; j += 256) {
      s_ext[2 * kMaskAxis + j] = pix_to_ndc(H - 1 - (min(H, j * 16 + 16) - 1), H, W);
      s_ext[3 * kMaskAxis + j] = pix_to_ndc(H - 1 - j * 16, H, W);
    }
  }
  if (zero_a) {
    float* z = zero_a + (int64_t)n * zero_a_n;
    for (int j = part * 256 + tid; j < zero_a_n; j += parts * 256) z[j] = 0.0f;
  }
  if (zero_b && n < M) {  // one texture gradient per MESH
    float* z = zero_b + (int64_t)n * zero_b_n;
    for (int j = part * 256 + tid; j < zero_b_n; j += parts * 256) z[j] = 0.0f;
  }
  __syncthreads();
  const int f = part * 256 + tid;
  if (f < F) {
    const float* rm = R + (int64_t)n * 9;
    const float* tv = T + (int64_t)n * 3;
    const int64_t* fc = faces + (int64_t)m * fstride + (int64_t)f * 3;
    const float* vp = verts + (int64_t)m * V * 3;
    float c[9];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int64_t vi = min(max(fc[k], (int64_t)0), (int64_t)V - 1);
      project_vertex(vp + vi * 3, rm, tv, sx, sy, c + 3 * k);
    }
    const int64_t i = (int64_t)n * F + f;
#pragma unroll
    for (int q = 0; q < 9; ++q) face_verts[i * 9 + q] = c[q];
    float4 bb, rc[5];
    const bool culled = face_record(c, r, cull, bb, rc);
    bbox[i] = bb;
    fz[i] = rc[4].w;
#pragma unroll
    for (int q = 0; q < 5; ++q) rec[5 * i + q] = rc[q];
    if (!culled) {
      int c0, c1, r0, r1;
      box_pixels(bb, H, W, c0, c1, r0, r1);
      for (int by = r0 >> 4; by <= (r1 >> 4); ++by)
        for (int bx = c0 >> 4; bx <= (c1 >> 4); ++bx) {
          if (masks) {
            if (!bbox_overlap(bb, s_ext[bx], s_ext[kMaskAxis + bx], s_ext[2 * kMaskAxis + by], s_ext[3 * kMaskAxis + by]))
              continue;
            atomicOr(&s_mask[(by * nbx + bx) * 8 + (tid >> 5)], 1u << (tid & 31));
          }
          atomicAdd(&s_load[by * nbx + bx], 1);
        }
    }
  }
  __syncthreads();
  int* out = binload + ((int64_t)n * parts + part) * nbins;
  for (int j = tid; j < nbins; j += 256) out[j] = s_load[j];
  if (masks)  // [mesh][bin][part][8 words]: the parts of a bin lie side by side
    for (int j = tid; j < nbins * 8; j += 256)
      masks[(((int64_t)n * nbins + (j >> 3)) * parts + part) * 8 + (j & 7)] = s_mask[j];
}

// ------------------------------------------------------------------------------------------------------------------
// rasterizer
// ------------------------------------------------------------------------------------------------------------------
struct RasterParams {
  const float4* bbox;
  const float4* rec;
  const int64_t* first;
  const float* fz;  // nearest corner depth per face
  const int64_t* nfaces;
  int N, H, W, K;
  int eqF;  // > 0: every mesh has eqF faces and mesh n starts at n * eqF (the one-call path) -- saves the kernels a
            // dependent load of first[n] / nfaces[n] before their first useful one
  float blur;
  int clip;
  int nbx, nby;  // bins per row / column
  int cap;       // K-list slots per lane held in LDS
  int KL;        // K-list capacity (list_cap(K) >= K)
  int64_t total_faces;
  const int* order;  // workgroup -> (mesh * bins + bin), heaviest bins first; kEmptyBin flags a bin no face box reaches
  const unsigned* masks;  // face masks of k_project_prep_mesh ([mesh][bin][part][8 words]) or null
  int parts;              // 256-face parts per mesh (masks only)
  unsigned long long* spill;  // [tiles][K-cap][64] overflow of the K-lists
  unsigned long long* full_thr;  // fused mode: [tiles][64] largest key of a pixel's list once it is full (see the pair route)
  // outputs (fragments)
  int32_t* p2f;
  float* zbuf;
  float* bary;
  float* dists;
  // tile-list export
  int32_t* tl_counts;
  int32_t* tl_lists;
  int tl_cap;
  // fused shading (MODE 2) and the fused backward
  const float* maps;
  const float* fuv;
  int fuv_shared;
  int Ht, Wt;
  float sigma;
  float bg[3];
  float amb[3];
  int use_amb;
  int mode;
  float* image;     // (N,4,H,W)
  int32_t* layers;  // per 64-pixel group K * 64 rows of kRowWords dwords (see kRowWords), then the per-pixel counts
  float* scratch;   // backward: one float per layer row (dL/dalpha) handed from the per-pixel sweep to the evenly
                    // dealt gradient pass
  const float* gimage;
  float* gmaps;
  float* gfv;
  // fused backward over equal-sized meshes (one-call path): sum_V > 0 makes the kernel add a face's corner gradients
  // straight into per-VERTEX sums (x, y of vertex v of mesh n at gfv[(n * sum_V + v) * 2 ...], the buffer being the
  // grad_face_verts allocation) through the faces table, so that the camera's backward has nothing left to gather
  int sum_V;
  const int64_t* faces;
  int64_t fstride;
  // copy-free pose tiling (ur_render_meshes_*): image n shows mesh n % M under pose n.  M == N (or 0) = one mesh per
  // image.  Textures, face tables and -- in the backward -- the texture gradient belong to the M meshes.
  int M;
};

// the mesh image n shows (see RasterParams::M)
__device__ __forceinline__ int mesh_of(const RasterParams& P, int n) { return (P.M > 0 && P.M < P.N) ? n % P.M : n; }

// face-UV table row of packed face `fid` of mesh n
__device__ __forceinline__ const float* face_uv_row(const RasterParams& P, int n, int fid) {
  int64_t fl = (int64_t)fid;
  if (P.fuv_shared) fl -= (P.eqF > 0 ? (int64_t)n * P.eqF : P.first[n]);
  else if (P.M > 0 && P.M < P.N) fl -= (int64_t)(n - n % P.M) * P.eqF;  // tiled poses: the row of mesh n % M
  return P.fuv + fl * 6;
}

constexpr int kScanQ = 8;         // faces per lane per scan round
constexpr int kBinCap = 256 * kScanQ;  // faces per scan round = bin-list entries (u16 offsets into the round)
constexpr int kSub = 64;         // faces staged in LDS per fine step
#ifndef UR_LDS_BUDGET
#define UR_LDS_BUDGET 40960
#endif
constexpr int kLdsBudget = UR_LDS_BUDGET;  // a quarter of the CU's 160 KB: four workgroups (16 waves) per CU (see raster_cap)
constexpr int kPreStride = 96;   // ints per wave in the pair-prefix array: 65 prefix entries, then 64 byte marks

// Capacity of a pixel's candidate list while faces are being collected.  The K nearest of all candidates are among the
// KL nearest for any KL >= K, so the lists may run longer than K and are cut to K after the final sort -- exact, and
// a list reaches its capacity (where every further candidate costs a scan of the whole list) far less often than it
// reaches K.  Up to 64 (128) entries a list is still sorted by one wave-wide network.
__host__ __device__ inline int list_cap(int K) { return K <= 64 ? (2 * K < 64 ? 2 * K : 64) : (K <= 128 ? 128 : K); }

// `layers`: per 64-pixel wave group up to K * 64 layer ROWS of 8 dwords (32 bytes = one memory sector, written and read
// as two dwordx4), in the group's pixel-major item order (the layers of pixel 0 front to back, then pixel 1, ...: row
// index = exclusive prefix of the per-pixel layer counts + layer), followed by the per-pixel layer counts of all
// groups.  Row = { packed face index, pixel (lane) of the group, transmittance in front of the layer, alpha,
//                  r, g, b, 0 }   (float bits)
constexpr int kRowWords = 8;
// where the per-pixel layer counts start inside `layers` (after the [group][K * 64] rows)
__host__ __device__ inline int64_t layer_counts_offset(int N, int nbx, int nby, int K) {
  return (int64_t)N * nbx * nby * 4 * K * 64 * kRowWords;
}

// A key packs (depth bits, face index, hides-the-rest flag): unsigned order == (depth, face) order (depths are >= +0).
__device__ __forceinline__ int key_face(unsigned long long k) { return (int)(((unsigned)(k & 0xffffffffull)) >> 1); }

// The per-lane top-K list: the first `cap` slots live in LDS ([slot][lane], 8-byte lane stride: conflict free), the
// rest (K > cap only, rare to be touched) in a global spill area of the same shape.
struct KeyList {
  unsigned long long* lds;    // + lane
  unsigned long long* spill;  // + lane
  int cap;
  int stride;                 // pixels per group: entries of one slot lie `stride` keys apart
  __device__ __forceinline__ unsigned long long get(int i) const {
    return i < cap ? lds[(size_t)i * stride] : spill[(size_t)(i - cap) * stride];
  }
  __device__ __forceinline__ void put(int i, unsigned long long v) const {
    if (i < cap) lds[(size_t)i * stride] = v; else spill[(size_t)(i - cap) * stride] = v;
  }
  // Keep, in order, the entries <= bound of this lane's first n entries (lanes with !need keep everything and
  // return n); mx = the largest n among the lanes that need it (wave-uniform).  Entries are read eight at a time
  // (LDS or spill alike), so that the loads overlap instead of forming one dependent chain; a write position never
  // passes the entries still to be read.
  __device__ __forceinline__ int compact(bool need, int n, int mx, unsigned long long bound) const {
    if (!need) n = 0;  // nothing to do in the loop below
    int j = 0;
    for (int i0 = 0; i0 < mx; i0 += 8) {
      unsigned long long t[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) t[q] = (i0 + q < n) ? get(i0 + q) : ~0ull;
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (i0 + q < n && t[q] <= bound) { put(j, t[q]); ++j; }
    }
    return j;
  }
};

// Fused mode, lists that are FULL (KL entries): the lanes in `pending` each hold a hit (key, pixel) that found its
// pixel's list full.  One at a time, all 64 lanes scan that list for its two largest entries; the newcomer replaces the
// largest if it beats it (the exact keep-the-KL-smallest rule, whatever the order of arrival), and the list's new
// largest key is left in thr[pixel] so that later candidates that cannot improve the list are dropped unseen.
__device__ __attribute__((noinline)) void full_list_insert(unsigned long long pending, unsigned long long over_key,
                                                           int over_pl, unsigned long long* lds_keys,
                                                           unsigned long long* spill_keys, unsigned long long* thr,
                                                           int cap, int KL, int stride) {
  const int lane = threadIdx.x & 63;
  while (pending) {
    wave_lds_handoff();  // lane 0's replacement of the last round is part of the list the next round scans
    const int srcl = __builtin_ctzll(pending);
    pending &= pending - 1;
    const int opl = __builtin_amdgcn_readlane(over_pl, srcl);
    const unsigned long long okey =
        ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(over_key >> 32), srcl) << 32) |
        (unsigned)__builtin_amdgcn_readlane((int)(unsigned)over_key, srcl);
    unsigned long long m1 = 0ull, m2 = 0ull;
    int i1 = 0;
    for (int j = lane; j < KL; j += 64) {
      const unsigned long long v = j < cap ? lds_keys[(size_t)j * stride + opl] : spill_keys[(size_t)(j - cap) * stride + opl];
      if (v > m1) { m2 = m1; m1 = v; i1 = j; }
      else if (v > m2) m2 = v;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {  // the wave's two largest (keys are unique)
      const unsigned long long o1 = __shfl_xor(m1, d, 64), o2 = __shfl_xor(m2, d, 64);
      const int oi = __shfl_xor(i1, d, 64);
      const unsigned long long lo = o1 < m1 ? o1 : m1;  // the smaller of the two maxima
      if (o1 > m1) { m1 = o1; i1 = oi; }
      m2 = m2 > o2 ? m2 : o2;
      m2 = m2 > lo ? m2 : lo;
    }
    if (lane == 0) {
      unsigned long long largest = m1;
      if (okey < m1) {
        if (i1 < cap) lds_keys[(size_t)i1 * stride + opl] = okey;
        else spill_keys[(size_t)(i1 - cap) * stride + opl] = okey;
        largest = okey > m2 ? okey : m2;
      }
      thr[opl] = largest;
    }
  }
}

// MODE 0: write fragments ; MODE 1: export tile lists ; MODE 2: fused UV sampling + layered blend (image + layers)
// BIGK (list capacity > 64): the code for lists longer than a wave and for scanning full lists that sit mostly in the spill area
// -- kept out of the K <= 64 build, whose register allocation it would disturb (1 % on the headline workload)
// MASKS (fused mode, equal-sized meshes): the bin's face list comes from the face masks the prep kernel left behind
// instead of a scan of all the mesh's face boxes: same test, same faces, same order -- 32 bytes per 256 faces to read
// instead of 256 boxes.
// G8 (fused mode): EIGHT waves per 16x16 bin, each owning 32 pixels (every second column, every fourth row) instead of
// four waves of 64.  A wave's chain of dependent steps -- pair batches, sorts, shading rounds, all in proportion to its
// pixels -- is half as long, and a launch has twice as many waves to fill the machine with (the reference's own 64x64
// training images are 2048 waves of 64 pixels for 4096 resident slots; at the headline size the kernel lasted as long
// as its heaviest bins' chains).  Workgroups of 512 threads, two per CU: the same 16 waves per CU.  Lanes 32..63 of
// such a wave own no pixel; they work wherever work is dealt over the lanes (pairs, items, sorting networks).
template <int MODE, bool BIGK, bool MASKS, bool G8 = false>
#ifndef UR_RASTER_WPS
#define UR_RASTER_WPS 4  // waves per SIMD the register budget is set for
#endif
__global__ void __launch_bounds__(G8 ? 512 : 256, UR_RASTER_WPS) k_raster(const RasterParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  static_assert(!G8 || MODE == 2, "eight-wave bins are the fused mode's");
  constexpr int BW = 16, BH = 16, WPB = G8 ? 8 : 4;
  constexpr int GP = 256 / WPB;     // pixels per wave group
  constexpr int ROWS = GP / 8;      // pixel rows of a group (8 columns each)
  constexpr int NT = 64 * WPB;      // threads per workgroup
  // LDS carve (raster_fixed_lds() bytes, then the keys):
  //   [bin list 2048 u16][wave counts 4 x WPB i32][per wave: 96 i32 pair prefix + marks (reused as the item prefix of
  //   the fused tail)][per wave: 64 u32 pair decode info][per wave: 64 i32 list lengths][per wave: 64 u64 hide keys]
  //   [staged bbox 2x64 float4][staged records 2x320 float4][G8: 512 float4 more for the tail's result slots][keys]
  uint16_t* s_bin = reinterpret_cast<uint16_t*>(smem);
  int32_t* s_wcnt = reinterpret_cast<int32_t*>(s_bin + kBinCap);
  int32_t* s_fpre_all = s_wcnt + 4 * WPB;
  uint32_t* s_finfo_all = reinterpret_cast<uint32_t*>(s_fpre_all + WPB * kPreStride);
  int32_t* s_cnt_all = reinterpret_cast<int32_t*>(s_finfo_all + WPB * 64);
  unsigned long long* s_hide_all = reinterpret_cast<unsigned long long*>(s_cnt_all + WPB * 64);
  float4* s_bb = reinterpret_cast<float4*>(s_hide_all + WPB * 64);
  float4* s_rec = s_bb + 2 * kSub;
  unsigned long long* s_keys_all = reinterpret_cast<unsigned long long*>(s_rec + 2 * kSub * 5 + (G8 ? 512 : 0));

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // workgroup schedule: heaviest bins first (k_bin_order)
  const int sched = P.order[blockIdx.x];
  const int item = sched & kBinIndex;
  const int n = item / (P.nbx * P.nby);
  const int b = item - n * (P.nbx * P.nby);
  const int by = b / P.nbx, bx = b - by * P.nbx;
  const int H = P.H, W = P.W, K = P.K, KL = P.KL;

  if (MODE == 2 && (sched & kEmptyBin)) {
    // a bin the schedule flagged empty (no face box reaches it; half the bins of a batch of centred objects): its
    // pixels are background and its layer counts zero -- written straight away, none of the set-up below
    const int ex = bx * BW + (wave & 1) + 2 * (lane & 7), ey = by * BH + (wave >> 1) + (G8 ? 4 : 2) * (lane >> 3);
    if (bx * BW + (wave & 1) < W && by * BH + (wave >> 1) < H && lane < GP) {
      const int64_t etile = (((int64_t)n * P.nby + by) * P.nbx + bx) * WPB + wave;
      P.layers[layer_counts_offset(P.N, P.nbx, P.nby, K) + etile * GP + lane] = 0;
      if (ex < W && ey < H) {
        const int64_t hw = (int64_t)H * W;
        float* img = P.image + (int64_t)n * 4 * hw + (int64_t)ey * W + ex;
        img[0] = 0.0f + 1.0f * 1.0f * P.bg[0];
        img[hw] = 0.0f + 1.0f * 1.0f * P.bg[1];
        img[2 * hw] = 0.0f + 1.0f * 1.0f * P.bg[2];
        img[3 * hw] = 1.0f - 1.0f;
      }
    }
    return;
  }

  // bin extents in pixel-centre NDC (columns x0..x1, rows y0..y1, clamped to the image)
  const int bx0 = bx * BW, by0 = by * BH;
  const int bx1 = min(W, bx0 + BW) - 1, by1 = min(H, by0 + BH) - 1;
  const NdcAxis ax_x = ndc_axis(W, H), ax_y = ndc_axis(H, W);
  const float bxc_hi = pix_to_ndc(W - 1 - bx0, ax_x), bxc_lo = pix_to_ndc(W - 1 - bx1, ax_x);
  const float byc_hi = pix_to_ndc(H - 1 - by0, ax_y), byc_lo = pix_to_ndc(H - 1 - by1, ax_y);

  // this wave's 64 pixels.  MODE 1 (tile lists): the 8x8 tile (wave & 1, wave >> 1) of the bin.  Otherwise every
  // second column / row of the bin starting at (wave & 1, wave >> 1): the four waves of a workgroup then see almost
  // the same load whatever the bin holds (a silhouette through one corner would otherwise leave three waves idle).
  // (G8: every second column and every FOURTH row from (wave & 1, wave >> 1), 8 columns x 4 rows)
  const int xs = (MODE == 1) ? 1 : 2;                 // pixel stride inside the wave: columns
  const int ys = (MODE == 1) ? 1 : (G8 ? 4 : 2);      //                               rows
  const int tx0 = bx0 + (wave & 1) * (MODE == 1 ? kTile : 1), ty0 = by0 + (wave >> 1) * (MODE == 1 ? kTile : 1);
  const bool tile_live = (tx0 < W) && (ty0 < H);
  const int tx1 = min(W, tx0 + xs * (kTile - 1) + 1) - 1, ty1 = min(H, ty0 + ys * (ROWS - 1) + 1) - 1;
  const float txc_hi = pix_to_ndc(W - 1 - tx0, ax_x), txc_lo = pix_to_ndc(W - 1 - max(tx1, tx0), ax_x);
  const float tyc_hi = pix_to_ndc(H - 1 - ty0, ax_y), tyc_lo = pix_to_ndc(H - 1 - max(ty1, ty0), ax_y);

  // this lane's pixel
  const int x = tx0 + xs * (lane & 7), y = ty0 + ys * (lane >> 3);
  const bool pix_live = (x < W) && (y < H) && (lane < GP);
  const float px = pix_to_ndc(W - 1 - x, ax_x), py = pix_to_ndc(H - 1 - y, ax_y);

  // index of this wave's pixel group in layers / spill / tile lists
  const int64_t tile_id = (MODE == 1) ? ((int64_t)n * ((H + 7) / 8) + (ty0 >> 3)) * ((W + 7) / 8) + (tx0 >> 3)
                                      : (((int64_t)n * P.nby + by) * P.nbx + bx) * WPB + wave;
  KeyList keys;
  keys.cap = P.cap;
  keys.stride = GP;
  keys.lds = s_keys_all + (size_t)wave * P.cap * GP + (lane & (GP - 1));
  keys.spill = P.spill + (tile_live ? tile_id : 0) * (int64_t)(KL - P.cap) * GP + (lane & (GP - 1));
  int cnt = 0;       // entries in this lane's list
  // BIGK, list full: the two largest entries found by the last scan of the list (full_known = how many of them still
  // hold: 2 after a scan, 1 after one replacement, 0 = scan again); any compaction or append resets it
  unsigned long long full_m1 = 0, full_m2 = 0;
  int full_i1 = 0, full_i2 = 0, full_known = 0;
  int tl_count = 0;  // MODE 1: faces appended to this tile's list (wave-uniform)
  // MODE 2 only: depth of the nearest hit found so far that hides everything behind it (alpha == 1 and
  // d(alpha)/d(dist) == 0).  A face whose nearest corner lies behind it cannot matter to this pixel's blend: its
  // depth at the pixel is a convex combination of its corner depths (the clipped barycentrics are >= 0 and sum to 1).
  float z_hide = __builtin_inff();
  unsigned long long hide_key = ~0ull;  // smallest key among the hits that hide the rest
  bool stale = false;                   // hide_key moved since this lane's list was last compacted
  const bool can_cull = (MODE == 2) && (P.clip != 0 || P.blur == 0.0f);
  // hides_rest() as two compares against wave-uniform bounds (the mode is a runtime value: testing it per pair costs
  // a tree of scalar branches): hard blend d <= 0 ; clip-inside d < 0 ; sigmoid d < -20 sigma
  const float hide_lt = P.mode == 2 ? -20.0f * P.sigma : (P.mode == 1 ? 0.0f : -__builtin_inff());
  const float hide_le = P.mode == 0 ? 0.0f : -__builtin_inff();

  // MODE 2 evaluates (face, pixel) PAIRS dealt out over the lanes (see the fine step); the per-pixel list length and
  // hide key then live in LDS, where any lane can reach them with integer atomics.
  int32_t* s_fpre = s_fpre_all + wave * kPreStride;
  uint32_t* s_finfo = s_finfo_all + wave * 64;
  int32_t* s_cnt = s_cnt_all + wave * 64;
  unsigned long long* s_hide = s_hide_all + wave * 64;
  unsigned long long* const lds_keys = s_keys_all + (size_t)wave * P.cap * GP;
  unsigned long long* const spill_keys = P.spill + (tile_live ? tile_id : 0) * (int64_t)(KL - P.cap) * GP;
  float pxc[8], pyc[8];                // the tile's 8 column / 8 row centres (wave-uniform)
  unsigned livecols = 0, liverows = 0;
  // a full list's largest key (all ones while the list is not full): candidates that do not beat it are dropped
  // without a look at the list
  unsigned long long* const thr_keys = (MODE == 2) ? P.full_thr + (tile_live ? tile_id : 0) * GP : nullptr;
  if (MODE == 2) {
    s_cnt[lane] = 0;
    s_hide[lane] = ~0ull;
    if (tile_live && lane < GP) thr_keys[lane] = ~0ull;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      pxc[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(px), j));
      pyc[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(py), j * 8));
      livecols |= (tx0 + xs * j < W) ? (1u << j) : 0u;
      liverows |= (j < ROWS && ty0 + ys * j < H) ? (1u << j) : 0u;
    }
  }

  const int64_t f0 = P.eqF > 0 ? (int64_t)n * P.eqF : P.first[n];
  // a bin flagged empty by the schedule has no faces to look at: straight to the output (background / padding)
  const int nf = (sched & kEmptyBin) ? 0 : (P.eqF > 0 ? P.eqF : (int)P.nfaces[n]);
  const int uwave = __builtin_amdgcn_readfirstlane(wave);

  for (int rbase = 0; rbase < nf; rbase += kBinCap) {
    // ---- coarse: the round's faces (up to 2048) are split evenly over the waves, wave w scans a contiguous range
    //      (independent bbox loads per lane).  The survivors are packed into one bin list, nearer half first (by
    //      nearest-corner depth against the midpoint of the bin's depth range), ascending face index inside each
    //      half: the fused blend can then skip most of the far side of a closed mesh (z_hide below).  Order never
    //      changes results -- selection is by (depth, face).
    const int fbase = (int32_t)f0 + rbase;  // packed index of the round's first face: face = fbase + s_bin[i]
    const int per = ((min(nf - rbase, kBinCap) + NT - 1) / NT) << 6;  // faces per wave (multiple of 64)
    const int wbase = rbase + wave * per;
    float zv[kScanQ];
    unsigned long long pm[kScanQ];
    int wc = 0;
    float zlo = __builtin_inff(), zhi = -__builtin_inff();
    if constexpr (MASKS) {
      // the round's 2048 mask bits: 64 words, one per lane (eight consecutive parts of this bin)
      const int part0 = rbase >> 8;
      const int nw = min(8, P.parts - part0) * 8;
      const unsigned* mp = P.masks + (((int64_t)n * (P.nbx * P.nby) + b) * P.parts + part0) * 8;
      const unsigned mw = lane < nw ? mp[lane] : 0u;
      if (__ballot(mw != 0u) == 0ull) continue;  // every wave reads the same words: a uniform decision
#pragma unroll
      for (int q = 0; q < kScanQ; ++q) {
        pm[q] = 0ull;
        zv[q] = 0.0f;
        if (q * 64 < per) {
          const int wi = (uwave * per + q * 64) >> 5;
          pm[q] = (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)mw, wi) |
                  ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)mw, wi + 1) << 32);
          if ((pm[q] >> lane) & 1ull) zv[q] = P.fz[f0 + wbase + q * 64 + lane];
        }
      }
#pragma unroll
      for (int q = 0; q < kScanQ; ++q) {
        wc += __popcll(pm[q]);
        if ((pm[q] >> lane) & 1ull) { zlo = min_s(zlo, zv[q]); zhi = max_s(zhi, zv[q]); }
      }
    } else {
      float4 bbv[kScanQ];
#pragma unroll
      for (int q = 0; q < kScanQ; ++q) {
        const int f = wbase + q * 64 + lane;
        const bool in = (q * 64 < per) && (f < nf);
        bbv[q] = in ? P.bbox[f0 + f] : make_float4(__builtin_inff(), -__builtin_inff(), 0.0f, 0.0f);
        zv[q] = (MODE == 2 && in) ? P.fz[f0 + f] : 0.0f;
      }
#pragma unroll
      for (int q = 0; q < kScanQ; ++q) {
        const bool pass = bbox_overlap(bbv[q], bxc_lo, bxc_hi, byc_lo, byc_hi);
        pm[q] = __ballot(pass);
        wc += __popcll(pm[q]);
        if (MODE == 2 && pass) { zlo = min_s(zlo, zv[q]); zhi = max_s(zhi, zv[q]); }
      }
    }
    if (MODE == 2) {
      zlo = wave_fmin(zlo);
      zhi = wave_fmax(zhi);
    }
    if (lane == 0) {
      s_wcnt[wave] = wc;
      s_wcnt[2 * WPB + wave] = __float_as_int(zlo);
      s_wcnt[3 * WPB + wave] = __float_as_int(zhi);
    }
    __syncthreads();
    float zmid = __builtin_inff();
    if (MODE == 2) {
      float a = __builtin_inff(), b2 = -__builtin_inff();
#pragma unroll
      for (int w = 0; w < WPB; ++w) {
        a = fminf(a, __int_as_float(s_wcnt[2 * WPB + w]));
        b2 = fmaxf(b2, __int_as_float(s_wcnt[3 * WPB + w]));
      }
      zmid = 0.5f * (a + b2);
    }
    unsigned long long nm[kScanQ];  // passing faces of the near half
    int wn = 0;
#pragma unroll
    for (int q = 0; q < kScanQ; ++q) {
      nm[q] = (MODE == 2) ? __ballot(((pm[q] >> lane) & 1ull) && zv[q] <= zmid) : pm[q];
      wn += __popcll(nm[q]);
    }
    if (lane == 0) s_wcnt[WPB + wave] = wn;
    __syncthreads();
    int list_n = 0, near_n = 0, pre_all = 0, pre_near = 0;
#pragma unroll
    for (int w = 0; w < WPB; ++w) {
      const int c = s_wcnt[w], cn = s_wcnt[WPB + w];
      pre_all += (w < wave) ? c : 0;
      pre_near += (w < wave) ? cn : 0;
      list_n += c;
      near_n += cn;
    }
    {
      int pos_near = pre_near, pos_far = near_n + (pre_all - pre_near);
      const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
      for (int q = 0; q < kScanQ; ++q) {
        const unsigned long long fm = pm[q] & ~nm[q];
        const uint16_t fid = (uint16_t)(wave * per + q * 64 + lane);
        if ((nm[q] >> lane) & 1ull) s_bin[pos_near + __popcll(nm[q] & lt)] = fid;
        else if ((fm >> lane) & 1ull) s_bin[pos_far + __popcll(fm & lt)] = fid;
        pos_near += __popcll(nm[q]);
        pos_far += __popcll(fm);
      }
    }
    __syncthreads();
    if (list_n == 0) continue;

    // ---- fine: 64 bin faces at a time; their bboxes / records are staged in LDS by the whole workgroup (next step's
    //      global loads are in flight while this step is evaluated), each wave walks the ones that touch its tile
    const int nsub = (list_n + kSub - 1) / kSub;
    float4 r_bb = make_float4(0, 0, 0, 0), r_r0 = r_bb, r_r1 = r_bb;
    auto stage_load = [&](int c) {
      const int cntc = min(kSub, list_n - c * kSub);
      if (tid < cntc) r_bb = P.bbox[fbase + s_bin[c * kSub + tid]];
      if (tid < cntc * 5) r_r0 = P.rec[5 * (size_t)(fbase + s_bin[c * kSub + tid / 5]) + tid % 5];
      if (NT < kSub * 5 && tid + 256 < cntc * 5)
        r_r1 = P.rec[5 * (size_t)(fbase + s_bin[c * kSub + (tid + 256) / 5]) + (tid + 256) % 5];
    };
    auto stage_write = [&](int c) {
      const int cntc = min(kSub, list_n - c * kSub);
      float4* bb = s_bb + (c & 1) * kSub;
      float4* rc = s_rec + (c & 1) * kSub * 5;
      if (tid < cntc) bb[tid] = r_bb;
      if (tid < cntc * 5) rc[tid] = r_r0;
      if (NT < kSub * 5 && tid + 256 < cntc * 5) rc[tid + 256] = r_r1;
    };
    stage_load(0);
    stage_write(0);
    __syncthreads();
    // one face of the staged chunk against this lane's own pixel (lane = pixel); list state in registers
    auto lockstep_face = [&](int c, int src, const float4* bbs, const float4* rcs) {
      const int ufid = fbase + s_bin[c * kSub + src];  // wave-uniform
      const float4 ubb = bbs[src];
      const Face fc = unpack_face(rcs[src * 5 + 0], rcs[src * 5 + 1], rcs[src * 5 + 2], rcs[src * 5 + 3],
                                  rcs[src * 5 + 4]);
      bool hit = pix_live && !(px < ubb.x || px > ubb.y || py < ubb.z || py > ubb.w);
      if (can_cull) hit = hit && !(fc.zmin * 0.9999f > z_hide);
      Eval ev;
      if (hit) hit = eval_face(fc, px, py, P.blur, P.clip != 0, ev);
      if (hit) {
        const bool hides = (MODE == 2) && hides_rest(ev.sd, P.sigma, P.mode);
        const unsigned long long key = ((unsigned long long)__float_as_uint(ev.pz) << 32) |
                                       ((unsigned)ufid << 1) | (hides ? 1u : 0u);
        if (hides && key < hide_key) { z_hide = ev.pz; hide_key = key; stale = true; }
        if (MODE == 2 && stale && cnt >= P.cap && key <= hide_key) {
          // about to leave LDS (or full): first throw out what the nearest hiding hit has made irrelevant
          int j = 0;
          for (int i = 0; i < cnt; ++i) {
            const unsigned long long kk = keys.get(i);
            if (kk <= hide_key) { keys.put(j, kk); ++j; }
          }
          cnt = j;
          stale = false;
          full_known = 0;
        }
        if (MODE == 2 && key > hide_key) {
          // behind a hit that hides the rest: irrelevant to the blend
        } else if (cnt < KL) {
          keys.put(cnt, key);
          ++cnt;
          full_known = 0;
        } else {
          // list full: the newcomer replaces the current maximum when it is smaller
          if constexpr (BIGK) {
            // Most of a long list sits in the global spill area, so a scan is dear (eight entries in flight at a
            // time) and full lists come in numbers (silhouette bins of dense meshes): a scan keeps the TWO largest
            // entries, which settles every newcomer that is no improvement and one replacement without another scan.
            if (full_known == 0) {
              unsigned long long m1 = 0, m2 = 0;
              int i1 = 0, i2 = 0;
              for (int i0 = 0; i0 < KL; i0 += 8) {
                unsigned long long t[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) t[q] = (i0 + q < KL) ? keys.get(i0 + q) : 0ull;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                  if (t[q] > m1) { m2 = m1; i2 = i1; m1 = t[q]; i1 = i0 + q; }
                  else if (t[q] > m2) { m2 = t[q]; i2 = i0 + q; }
                }
              }
              full_m1 = m1; full_i1 = i1; full_m2 = m2; full_i2 = i2;
              full_known = 2;
            }
            if (key < full_m1) {
              keys.put(full_i1, key);
              if (full_known == 2) {  // the new maximum: the runner-up, or the newcomer where it sits
                if (key > full_m2) full_m1 = key;
                else { full_m1 = full_m2; full_i1 = full_i2; }
                full_known = 1;
              } else {
                full_known = 0;
              }
            }
          } else {
            unsigned long long mx = 0;
            int im = 0;
            for (int i0 = 0; i0 < KL; i0 += 4) {
              unsigned long long t[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) t[q] = (i0 + q < KL) ? keys.get(i0 + q) : 0ull;
#pragma unroll
              for (int q = 0; q < 4; ++q)
                if (t[q] > mx) { mx = t[q]; im = i0 + q; }
            }
            if (key < mx) keys.put(im, key);
          }
        }
      }
    };

    for (int c = 0; c < nsub; ++c) {
      if (c + 1 < nsub) stage_load(c + 1);
      if (tile_live) {
        const int cntc = min(kSub, list_n - c * kSub);
        const float4* bbs = s_bb + (c & 1) * kSub;
        const float4* rcs = s_rec + (c & 1) * kSub * 5;
        float4 mybb = make_float4(__builtin_inff(), -__builtin_inff(), 0.0f, 0.0f);
        if (lane < cntc) mybb = bbs[lane];
        const bool pass = (lane < cntc) && bbox_overlap(mybb, txc_lo, txc_hi, tyc_lo, tyc_hi);
        if (MODE == 1) {
          const unsigned long long m = __ballot(pass);
          if (pass) {
            const int pos = tl_count + __popcll(m & ((1ull << lane) - 1ull));
            if (pos < P.tl_cap) P.tl_lists[tile_id * P.tl_cap + pos] = fbase + s_bin[c * kSub + lane];
          }
          tl_count += __popcll(m);
        } else if (MODE == 0) {
          unsigned long long m = __ballot(pass);
          while (m) {
            const int src = __builtin_ctzll(m);
            m &= m - 1;
            lockstep_face(c, src, bbs, rcs);
          }
        } else {
          // ---- MODE 2: a face of this mesh size covers a third of a tile's pixels, so walking faces with lane =
          //      pixel leaves two lanes in three idle.  Instead every lane first takes one FACE of the chunk and
          //      finds the rectangle of tile pixels inside its blur-expanded box (the very comparisons a pixel would
          //      make, on the tile's 8 column / 8 row centres); the (face, pixel) pairs of all rectangles are then
          //      numbered by a prefix sum and evaluated 64 at a time, whole faces per batch.  A hit is appended to
          //      its pixel's list with an LDS integer atomic; list order is irrelevant (sorted by key in the tail).
          unsigned colm = 0, rowm = 0;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            colm |= !(pxc[j] < mybb.x || pxc[j] > mybb.y) ? (1u << j) : 0u;
            rowm |= !(pyc[j] < mybb.z || pyc[j] > mybb.w) ? (1u << j) : 0u;
          }
          colm &= livecols;
          rowm &= liverows;
          const int rw = __popc(colm), rh = __popc(rowm);
          // Depth cull of whole faces: a face whose nearest corner lies behind the deepest hiding hit found so far at
          // any pixel of its rectangle -- every pixel of the rectangle having one -- would be rejected pair by pair
          // below; it gets no pairs at all.  (Snapshot taken once per chunk: hide depths only come nearer, pixels only
          // become covered, so a stale snapshot errs on the safe side.)
          bool behind = false;
          if (can_cull) {
            const unsigned long long hk0 = __hip_atomic_load(&s_hide[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            const unsigned long long uncovered = __ballot(hk0 == ~0ull);
            const int zb = wave_max(hk0 == ~0ull ? 0 : (int)(hk0 >> 32));  // depths are >= +0: bits order like ints
            unsigned long long rows = (unsigned long long)rowm * 0x0101010101010101ull & 0x8040201008040201ull;
            rows = (((rows + 0x7f7f7f7f7f7f7f7full) | rows) & 0x8080808080808080ull) >> 7;  // byte r = 1 iff row r
            const unsigned long long rect = (rows * 0xffull) & ((unsigned long long)colm * 0x0101010101010101ull);
            const float fzmin = (lane < cntc) ? rcs[lane * 5 + 4].w : 0.0f;
            behind = (rect & uncovered) == 0ull && fzmin * 0.9999f > __int_as_float(zb);
          }
          const int area = (pass && !behind) ? rw * rh : 0;
          const unsigned long long m = __ballot(area > 0);
          const int nfp = __popcll(m);
          if (nfp > 0) {
            const int incl = wave_scan_add(area);
            const int total_pairs = __builtin_amdgcn_readlane(incl, 63);
            wave_lds_handoff();  // (the last chunk's batches have read these slots)
            if (area > 0) {
              const int rank = __popcll(m & ((1ull << lane) - 1ull));
              s_fpre[rank] = incl - area;
              // decode word: chunk slot, rectangle origin, width and ceil(2^15 / width) (q / width for q < 64)
              s_finfo[rank] = (unsigned)lane | ((unsigned)(__ffs(colm) - 1) << 6) | ((unsigned)(__ffs(rowm) - 1) << 9) |
                              ((unsigned)rw << 12) | ((unsigned)((32768 + rw - 1) / rw) << 16);
            }
            if (lane + 1 >= nfp) s_fpre[lane + 1] = (lane + 1 == nfp) ? total_pairs : 0x7fffffff;
            wave_lds_handoff();
            // compacted view: lane r holds the r-th face that has pairs
            const int cpre = s_fpre[lane], cend = s_fpre[lane + 1];
            {
              const int r0 = 0;
              // Lists are not compacted while faces are collected: entries that a later, nearer hiding hit made
              // irrelevant are dropped in the tail's sorting passes, and a full list sheds them by itself (they are its
              // largest keys, so the newcomers that matter replace them one by one).  In-loop compactions cost the
              // dense-mesh workloads more than the shorter lists saved (5120-face meshes: +2.4 % without them).
              // Every face goes the pair way.  A hit on a FULL list (KL entries) is kept in its lane's registers
              // until the batch is through; the wave then takes those hits one by one, all 64 lanes scanning the
              // pixel's list together for its two largest entries: the newcomer replaces the largest if it beats it
              // -- the exact keep-the-KL-smallest rule, whatever the order -- and the list's new largest key is
              // remembered, so that later candidates that cannot improve the list cost nothing.
              const int r1 = nfp;
              const int pbeg = __builtin_amdgcn_readlane(cpre, r0), pend = __builtin_amdgcn_readlane(cend, r1 - 1);
              for (int pb = pbeg; pb < pend; pb += 64) {
                const int pidx = pb + lane;
                const bool act = pidx < pend;
                unsigned long long over_key = ~0ull;  // this lane's hit on a full list, if any (handled after the batch)
                // which face does pair pidx belong to?  The faces that start inside the batch mark their first
                // pair's slot with their rank + 1, slot 0 takes the face that holds pair pb, and a running maximum
                // spreads the marks to the right.
                unsigned char* marks = reinterpret_cast<unsigned char*>(s_fpre + 68);
                wave_lds_handoff();  // (the last batch has read its marks)
                marks[lane] = 0;
                const bool mine = lane >= r0 && lane < r1 && cend > pb && cpre < pb + 64;
                const int rfirst = __builtin_ctzll(__ballot(mine));
                if (mine && cpre > pb) marks[cpre - pb] = (unsigned char)(lane + 1);
                wave_lds_handoff();
                const int mark = (int)marks[lane];  // read by every lane: a select below, not a branch around the load
                const int rr = wave_scan_max(lane == 0 ? rfirst + 1 : mark) - 1;
                const int mypre = s_fpre[rr];
                const unsigned info = s_finfo[rr];
                const int q = pidx - mypre;
                const int w = (int)((info >> 12) & 15u);
                const int dy = (q * (int)(info >> 16)) >> 15, dx = q - dy * w;
                const int src = (int)(info & 63u);
                const int pl = act ? ((((int)((info >> 9) & 7u) + dy) << 3) | ((int)((info >> 6) & 7u) + dx)) : lane;
                const float qx = __shfl(px, pl, 64), qy = __shfl(py, pl, 64);
                if (act) {
                  const Face fc = unpack_face(rcs[src * 5 + 0], rcs[src * 5 + 1], rcs[src * 5 + 2], rcs[src * 5 + 3],
                                              rcs[src * 5 + 4]);
                  unsigned long long hk = __hip_atomic_load(&s_hide[pl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                  bool hit = true;
                  if (can_cull && hk != ~0ull) hit = !(fc.zmin * 0.9999f > __uint_as_float((unsigned)(hk >> 32)));
                  Eval ev;
                  if (hit) hit = eval_face(fc, qx, qy, P.blur, P.clip != 0, ev);
                  if (hit) {
                    const bool hides = ev.sd < hide_lt || ev.sd <= hide_le;  // = hides_rest(ev.sd, P.sigma, P.mode)
                    const unsigned long long key = ((unsigned long long)__float_as_uint(ev.pz) << 32) |
                                                   ((unsigned)(fbase + s_bin[c * kSub + src]) << 1) | (hides ? 1u : 0u);
                    if (hides && key < hk) { atomicMin(&s_hide[pl], key); hk = key; }
                    if (key <= hk) {
                      const int slot = atomicAdd(&s_cnt[pl], 1);
                      if (slot < KL) {
                        if (slot < P.cap) lds_keys[(size_t)slot * GP + pl] = key;
                        else spill_keys[(size_t)(slot - P.cap) * GP + pl] = key;
                      } else {
                        atomicSub(&s_cnt[pl], 1);  // the list is full
                        if (key < thr_keys[pl]) over_key = key;
                      }
                    }
                  }
                }
                // hits on full lists, one at a time, the whole wave on each (out of line: rare, and its registers would
                // otherwise count against the loop above)
                const unsigned long long pending = __ballot(over_key != ~0ull);
                if (pending) {
                  wave_lds_handoff();  // the lists it scans were appended to by any lane
                  full_list_insert(pending, over_key, pl, lds_keys, spill_keys, thr_keys, P.cap, KL, GP);
                  wave_lds_handoff();
                }
              }
            }
          }
        }
      }
      if (c + 1 < nsub) stage_write(c + 1);
      __syncthreads();
    }
  }

  if (MODE == 1) {
    if (tile_live) {
      if (lane == 0) P.tl_counts[tile_id] = tl_count;
      for (int i = tl_count + lane; i < P.tl_cap; i += 64) P.tl_lists[tile_id * P.tl_cap + i] = -1;
    }
    return;
  }
  if (!tile_live) return;

  if (MODE == 2) {
    // ---- fused shading.  (a) per pixel: drop the hits behind the nearest one that hides the rest, sort what is left
    //      front to back (in place; that hit, if any, comes last);  (b) the (pixel, layer) items of the whole tile
    //      are dealt out evenly over the 64 lanes -- any lane can evaluate any item -- so every lane stays busy
    //      although pixels have very different depths;  (c) each pixel pulls its items' (alpha, colour) back from
    //      the lanes that computed them, in depth order, and composites: the arithmetic and its order are exactly
    //      those of a per-pixel loop, so the image is bit-identical to the two-stage path.
    const float* map = P.maps + (int64_t)mesh_of(P, n) * P.Ht * P.Wt * 3;
    unsigned long long* lds_base = lds_keys;
    unsigned long long* spill_base = spill_keys;
    cnt = s_cnt[lane];
    hide_key = s_hide[lane];
    int n_rel = cnt;
    constexpr int kDeep = 8;
    // Lists too long for the register network (more than kDeep entries) first drop, in place, what their pixel's nearest
    // hiding hit made irrelevant (nothing was dropped while faces were collected): many of them then fit the register
    // network after all, and the cross-lane sorts get shorter (5120-face meshes +3 %, 20480 faces / K=100 +3 %).  Short
    // lists drop their hidden entries on the way through the network: they are replaced by the all-ones pad, which
    // sorts to the end.
    // (only when the wave holds many such lists: the drop is a lockstep loop over the longest of them, part of it
    // through the global spill area; a few long lists cost less going straight to the cross-lane sorts, which drop
    // hidden entries as well -- 128x128 -1 %, the 192-image pose step -3 %, dense meshes unchanged)
    if (__popcll(__ballot(hide_key != ~0ull && cnt > kDeep)) > 16) {
      const bool need = hide_key != ~0ull && cnt > kDeep;
      const int maxcnt = __builtin_amdgcn_readfirstlane(wave_max(need ? cnt : 0));
      const int j = keys.compact(need, cnt, maxcnt, hide_key);
      if (need) n_rel = j;
      wave_lds_handoff();  // the cross-lane sorts below read these lists
    }
    if (!pix_live) n_rel = 0;
    {
      // short lists (up to kDeep entries): loaded into registers, hidden entries replaced by the all-ones pad (no
      // real key has that value: depth bits of a finite float), sorted by a fixed compare-exchange network sized by
      // the longest short list of the wave, stored back.  Keys are unique, so the order is total.
      const bool small = n_rel <= kDeep;
      const int maxs = __builtin_amdgcn_readfirstlane(wave_max(small ? n_rel : 0));
      if (maxs > 0) {
        unsigned long long e[kDeep];
#pragma unroll
        for (int i = 0; i < kDeep; ++i) {
          e[i] = ~0ull;
          if (i < maxs) {
            if (small && i < n_rel) e[i] = keys.get(i);
            if (e[i] > hide_key) e[i] = ~0ull;
          }
        }
        auto cx = [&](int i, int j) {
          const bool sw = e[j] < e[i];
          const unsigned long long lo = sw ? e[j] : e[i], hi = sw ? e[i] : e[j];
          e[i] = lo; e[j] = hi;
        };
        if (maxs > 4) {
          cx(0, 1); cx(2, 3); cx(4, 5); cx(6, 7); cx(0, 2); cx(1, 3); cx(4, 6); cx(5, 7); cx(1, 2); cx(5, 6);
          cx(0, 4); cx(1, 5); cx(2, 6); cx(3, 7); cx(2, 4); cx(3, 5); cx(1, 2); cx(3, 4); cx(5, 6);
        } else if (maxs > 2) {
          cx(0, 1); cx(2, 3); cx(0, 2); cx(1, 3); cx(1, 2);
        } else if (maxs > 1) {
          cx(0, 1);
        }
        if (small) {
          int kept = 0;
#pragma unroll
          for (int i = 0; i < kDeep; ++i) kept += (e[i] != ~0ull) ? 1 : 0;
          n_rel = kept;
#pragma unroll
          for (int i = 0; i < kDeep; ++i)
            if (i < maxs && i < n_rel) keys.put(i, e[i]);
        }
      }
      if constexpr (BIGK) {
        // lists longer than 128 entries (K > 128 only): per-lane selection sort
        const bool mine = n_rel > 128;
        int sorted = 0;
        while (__any(mine && sorted < n_rel - 1)) {
          if (mine && sorted < n_rel - 1) {
            unsigned long long best = keys.get(sorted);
            const unsigned long long cur = best;
            int ib = sorted;
            for (int i = sorted + 1; i < n_rel; ++i) {
              const unsigned long long kk = keys.get(i);
              if (kk < best) { best = kk; ib = i; }
            }
            if (ib != sorted) { keys.put(ib, cur); keys.put(sorted, best); }
            ++sorted;
          }
        }
      }
      // long lists (pixels in the blur halo of a silhouette vertex, 9..64 entries): sorted across lanes by a bitonic
      // network over lane groups of 16, 32 or 64 -- four, two or one list per round, entry j of a list in lane j of
      // its group, padded with all-ones keys.  The lists of a round are fetched together (the part past the LDS cap
      // sits in the global spill area), so their round trips overlap.
      auto bitonic_pass = [&](auto WC, unsigned long long todo) {
        constexpr int Wd = decltype(WC)::value, per = 64 / Wd;
        const int sub = lane / Wd, j = lane & (Wd - 1);
        while (todo) {
          int myl = 0;
          bool have = false;
#pragma unroll
          for (int u = 0; u < per; ++u) {
            const bool any = todo != 0ull;
            const int l = any ? __builtin_ctzll(todo) : 0;
            todo &= todo - 1;
            if (sub == u) { myl = l; have = any; }
          }
          const int nl = __shfl(n_rel, myl, 64);  // every lane takes part: a masked-off lane cannot be read from
          const unsigned long long hk = __shfl(hide_key, myl, 64);
          const int myn = have ? nl : 0;
          unsigned long long v = ~0ull;
          if (j < myn) v = (j < P.cap) ? lds_base[(size_t)j * GP + myl] : spill_base[(size_t)(j - P.cap) * GP + myl];
          if (v > hk) v = ~0ull;  // hidden
#pragma unroll
          for (int k = 2; k <= Wd; k <<= 1)
#pragma unroll
            for (int d = k >> 1; d > 0; d >>= 1) {
              const unsigned long long o = lane_xor_d(v, d);
              const bool take_min = ((j & d) == 0) == ((j & k) == 0);
              v = (take_min == (o < v)) ? o : v;
            }
          const unsigned long long live = __ballot(v != ~0ull) >> (sub * Wd);
          const int kept = __popcll(Wd == 64 ? live : (live & ((1ull << (Wd & 63)) - 1ull)));
          if (j < kept) {
            if (j < P.cap) lds_base[(size_t)j * GP + myl] = v;
            else spill_base[(size_t)(j - P.cap) * GP + myl] = v;
          }
          if (have && j == 0) s_cnt[myl] = kept;  // the list's new length, picked up by its pixel below
        }
      };
      const unsigned long long d16 = __ballot(n_rel > kDeep && n_rel <= 16), d32 = __ballot(n_rel > 16 && n_rel <= 32),
                               d64 = __ballot(n_rel > 32 && n_rel <= 64);
      bitonic_pass(std::integral_constant<int, 16>{}, d16);
      bitonic_pass(std::integral_constant<int, 32>{}, d32);
      bitonic_pass(std::integral_constant<int, 64>{}, d64);
      if constexpr (BIGK) {
        // 65..128 entries (K > 64 only): one list at a time, two entries per lane (entry e in lane e & 63, register
        // e >> 6); the distance-64 exchange is between a lane's own two registers
        unsigned long long d128 = __ballot(n_rel > 64 && n_rel <= 128);
        while (d128) {
          const int l = __builtin_ctzll(d128);
          d128 &= d128 - 1;
          const int nl = __builtin_amdgcn_readlane(n_rel, l);
          const unsigned long long hk =
              ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(hide_key >> 32), l) << 32) |
              (unsigned)__builtin_amdgcn_readlane((int)(unsigned)hide_key, l);
          unsigned long long v[2];
#pragma unroll
          for (int r2 = 0; r2 < 2; ++r2) {
            const int e = lane + 64 * r2;
            v[r2] = ~0ull;
            if (e < nl) v[r2] = (e < P.cap) ? lds_base[(size_t)e * GP + l] : spill_base[(size_t)(e - P.cap) * GP + l];
            if (v[r2] > hk) v[r2] = ~0ull;  // hidden
          }
#pragma unroll
          for (int k = 2; k <= 128; k <<= 1)
#pragma unroll
            for (int d = k >> 1; d > 0; d >>= 1) {
              if (d == 64) {
                const bool sw = v[1] < v[0];
                const unsigned long long lo = sw ? v[1] : v[0], hi = sw ? v[0] : v[1];
                v[0] = lo; v[1] = hi;
              } else {
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2) {
                  const int e = lane + 64 * r2;
                  const unsigned long long o = lane_xor_d(v[r2], d);
                  const bool take_min = ((e & d) == 0) == ((e & k) == 0);
                  v[r2] = (take_min == (o < v[r2])) ? o : v[r2];
                }
              }
            }
#pragma unroll
          for (int r2 = 0; r2 < 2; ++r2) {
            const int e = lane + 64 * r2;
            if (v[r2] != ~0ull) {  // sorted: the kept entries come first
              if (e < P.cap) lds_base[(size_t)e * GP + l] = v[r2];
              else spill_base[(size_t)(e - P.cap) * GP + l] = v[r2];
            }
          }
          const int kept = __popcll(__ballot(v[0] != ~0ull)) + __popcll(__ballot(v[1] != ~0ull));
          if (lane == 0) s_cnt[l] = kept;
        }
      }
      // the pixels whose lists went through a cross-lane sort pick up their new lengths
      wave_lds_handoff();
      if (n_rel > kDeep && n_rel <= (BIGK ? 128 : 64)) n_rel = s_cnt[lane];
    }
    const int m = min(n_rel, K);  // layers of this pixel that matter, front to back in slots [0, m): the K nearest
    const int incl = wave_scan_add(m);
    const int pre = incl - m;
    const int total = __builtin_amdgcn_readlane(incl, 63);
    int* s_pre = s_fpre_all + wave * kPreStride;
    unsigned char* marks = reinterpret_cast<unsigned char*>(s_pre + 68);
    s_pre[lane] = pre;
    wave_lds_handoff();
    // the staging buffers are dead once the fine step is over (every wave is past its last barrier): this wave's
    // slice of them carries the batch's (alpha, r, g, b) from the lanes that computed them to the pixels' owners
    // (eight waves need 8 x 160 float4: the staged boxes, the staged records and the 512 float4 behind them)
    float4* s_res = (G8 ? s_bb : s_rec) + wave * 160;        // 128 float4: (alpha, r, g, b) of a round's items
    float* s_occ = reinterpret_cast<float*>(s_res + 128);    // 128 floats: transmittance in front of each of them
    int4* lay_rows = reinterpret_cast<int4*>(P.layers + tile_id * (int64_t)K * GP * kRowWords);  // [item][2 x int4]
    float occ = 1.0f, r = 0.0f, g = 0.0f, b = 0.0f;
    // One round takes 64 U items, U per lane.  The item phase is a chain of dependent memory round trips (key ->
    // record + UVs -> texels) with little arithmetic in between, so two items per lane are walked through it side by
    // side (every stage written for both before the next begins, no branches in between: lanes past the end redo the
    // last item and only their stores are masked), which lets their loads overlap.  Groups with at most 64 items
    // left take the one-item form.
    auto round = [&](auto UC, int base) {
      constexpr int U = decltype(UC)::value;
      int il[U], ik[U], ifid[U];
      bool act[U];
      // owner of an item: the pixels whose first item falls inside the 64-item batch mark that slot with lane + 1,
      // slot 0 takes the pixel that holds the batch's first item, and a running maximum spreads the marks to the right
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int b0 = base + 64 * u;
        wave_lds_handoff();  // (the marks of the previous pass have been read)
        marks[lane] = 0;
        const int l0 = 63 - __builtin_clzll(__ballot(m > 0 && pre <= b0));
        if (m > 0 && pre > b0 && pre < b0 + 64) marks[pre - b0] = (unsigned char)(lane + 1);
        wave_lds_handoff();
        il[u] = wave_scan_max(lane == 0 ? l0 + 1 : (int)marks[lane]) - 1;
        act[u] = b0 + lane < total;
        ik[u] = min(b0 + lane, total - 1) - s_pre[il[u]];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        unsigned long long key = lds_base[(size_t)min(ik[u], P.cap - 1) * GP + il[u]];
        if (ik[u] >= P.cap) key = spill_base[(size_t)(ik[u] - P.cap) * GP + il[u]];
        ifid[u] = key_face(key);
      }
      Face fc[U];
      float fu[U][6];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        fc[u] = load_face(P.rec, ifid[u]);
        const float* fp = face_uv_row(P, n, ifid[u]);
#pragma unroll
        for (int q = 0; q < 6; ++q) fu[u][q] = fp[q];
      }
      float va[U], vr[U], vg[U], vb[U];
      Bilin bl[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float qx = pix_to_ndc(W - 1 - (tx0 + xs * (il[u] & 7)), ax_x);
        const float qy = pix_to_ndc(H - 1 - (ty0 + ys * (il[u] >> 3)), ax_y);
        Eval ev;
        eval_face(fc[u], qx, qy, P.blur, P.clip != 0, ev);
        va[u] = layer_alpha(ev.sd, true, P.sigma, P.mode);
        const float tu = (ev.c0 * fu[u][0] + ev.c1 * fu[u][2]) + ev.c2 * fu[u][4];
        const float tv = (ev.c0 * fu[u][1] + ev.c1 * fu[u][3]) + ev.c2 * fu[u][5];
        bl[u] = bilin_setup(tu, tv, P.Ht, P.Wt);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) sample_rgb(map, bl[u], P.Ht, P.Wt, vr[u], vg[u], vb[u]);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (P.use_amb) { vr[u] *= P.amb[0]; vg[u] *= P.amb[1]; vb[u] *= P.amb[2]; }
        s_res[64 * u + lane] = make_float4(va[u], vr[u], vg[u], vb[u]);
      }
      wave_lds_handoff();
      // owners composite their items of this round, in depth order, and leave the transmittance in front of each
      const int s0 = max(pre, base), e0 = min(pre + m, base + 64 * U);
      const int nloop = __builtin_amdgcn_readfirstlane(wave_max(max(e0 - s0, 0)));
      // (four steps per trip, their LDS reads issued together: the deepest pixel of the round sets the trip count, up
      // to K, and every step used to wait for its own read; composite time of the last waves 4.4 -> 3.6 us, the
      // kernel -0.5 us at the headline size.  The arithmetic and its order are unchanged.)
#ifndef UR_COMP_UNROLL
#define UR_COMP_UNROLL 4
#endif
      for (int t = 0; t < nloop; t += UR_COMP_UNROLL) {
        float4 c[UR_COMP_UNROLL];
#pragma unroll
        for (int i = 0; i < UR_COMP_UNROLL; ++i) c[i] = s_res[min(max(s0 + t + i - base, 0), 64 * U - 1)];
#pragma unroll
        for (int i = 0; i < UR_COMP_UNROLL; ++i)
          if (s0 + t + i < e0) {
            s_occ[s0 + t + i - base] = occ;
            r += occ * c[i].x * c[i].y; g += occ * c[i].x * c[i].z; b += occ * c[i].x * c[i].w;
            occ = occ * (1.0f - c[i].x);
          }
      }
      // the layer rows for the backward, written by the lanes that computed them: 64 consecutive 32-byte rows per
      // store instruction
      wave_lds_handoff();
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (act[u]) {
          int4* row = lay_rows + (int64_t)(base + 64 * u + lane) * 2;
          row[0] = make_int4(ifid[u], il[u], __float_as_int(s_occ[64 * u + lane]), __float_as_int(va[u]));
          row[1] = make_int4(__float_as_int(vr[u]), __float_as_int(vg[u]), __float_as_int(vb[u]), 0);
        }
    };
    int base = 0;
    for (; base + 64 < total; base += 128) round(std::integral_constant<int, 2>{}, base);
    if (base < total) round(std::integral_constant<int, 1>{}, base);
    if (lane < GP) P.layers[layer_counts_offset(P.N, P.nbx, P.nby, K) + tile_id * GP + lane] = m;  // per-pixel layer count
    if (pix_live) {
      const int64_t hw = (int64_t)H * W;
      float* img = P.image + (int64_t)n * 4 * hw + (int64_t)y * W + x;
      img[0] = r + occ * 1.0f * P.bg[0];
      img[hw] = g + occ * 1.0f * P.bg[1];
      img[2 * hw] = b + occ * 1.0f * P.bg[2];
      img[3 * hw] = 1.0f - occ;
    }
    return;
  }

  // ---- output: ascending extraction, winners re-evaluated ----------------------------------------------------
  const int64_t ebase = (((int64_t)n * H + y) * W + x) * K;
  unsigned long long last_key = 0;
  bool first_take = true;
  const int nout = min(cnt, K);  // the list may hold more candidates than K (list_cap): the K nearest go out
  const int maxcnt = __builtin_amdgcn_readfirstlane(wave_max(nout));
  for (int k = 0; k < maxcnt; ++k) {
    if (k < nout) {
      unsigned long long best = ~0ull;
      for (int i = 0; i < cnt; ++i) {
        const unsigned long long kk = keys.get(i);
        if ((first_take || kk > last_key) && kk < best) best = kk;
      }
      last_key = best;
      first_take = false;
      const int fid = key_face(best);
      const Face fc = load_face(P.rec, fid);
      Eval ev;
      eval_face(fc, px, py, P.blur, P.clip != 0, ev);
      P.p2f[ebase + k] = fid;
      if (P.zbuf) P.zbuf[ebase + k] = __uint_as_float((unsigned)(best >> 32));
      if (P.dists) P.dists[ebase + k] = ev.sd;
      if (P.bary) {
        P.bary[(ebase + k) * 3 + 0] = ev.c0;
        P.bary[(ebase + k) * 3 + 1] = ev.c1;
        P.bary[(ebase + k) * 3 + 2] = ev.c2;
      }
    }
  }
  // padding: for every pixel of the tile the slots [nout, K) are written by lanes that sweep k (contiguous stores)
  for (int p = 0; p < 64; ++p) {
    const int pcnt = __builtin_amdgcn_readlane(nout, p);
    const int qx = tx0 + xs * (p & 7), qy = ty0 + ys * (p >> 3);
    if (qx >= W || qy >= H) continue;
    const int64_t qbase = (((int64_t)n * H + qy) * W + qx) * K;
    for (int k = pcnt + lane; k < K; k += 64) {
      P.p2f[qbase + k] = -1;
      if (P.zbuf) P.zbuf[qbase + k] = -1.0f;
      if (P.dists) P.dists[qbase + k] = -1.0f;
    }
    if (P.bary)
      for (int k3 = pcnt * 3 + lane; k3 < K * 3; k3 += 64) P.bary[qbase * 3 + k3] = -1.0f;
  }
}

// 64-pixel groups (one per wave) of an image: 4 per 16x16 bin
size_t wave_groups(int H, int W) { return (size_t)((H + 15) / 16) * ((W + 15) / 16) * 4; }

// LDS of k_raster before the key lists (see its carve); wpb = waves per bin (4, or 8 in the G8 build)
constexpr size_t raster_fixed_lds(int wpb) {
  return (size_t)kBinCap * 2 + (size_t)wpb * 16 + (size_t)wpb * kPreStride * 4 + (size_t)wpb * 64 * 4 +
         (size_t)wpb * 64 * 4 + (size_t)wpb * 64 * 8 + 2 * kSub * 16 + 2 * kSub * 5 * 16 + (wpb == 8 ? 512 * 16 : 0);
}

// Does the fused path run eight waves of 32 pixels per bin (k_raster<2, ., ., true>) rather than four of 64?  A pure
// function of the call's sizes: the forward and the backward of a step must agree on the layout of `layers`.
#ifndef UR_FUSED_G8
#define UR_FUSED_G8 1
#endif
// Measured (profiles/r03_ab_g8.txt, graph replay): it pays when the launch is too small to fill the chip with four
// waves per bin -- the reference's own training size, 32 images of 64x64: 2048 waves for 4096 resident slots, +16 % --
// and costs 13-20 % where the machine is full anyway (twice the per-chunk rectangle work, half-empty lane = pixel
// steps).  So: eight waves while all bins of the launch, four waves each, would not fill the resident wave slots.
constexpr int kResidentWaves = 256 * 16;  // MI355X: 256 CUs x 16 waves at this kernel's register budget
bool fused_g8(int N, int H, int W, int K) {
  (void)K;
  if (UR_FUSED_G8 == 0) return false;
  if (UR_FUSED_G8 == 2) return true;  // (build switch for A/B timing: always)
  return (int64_t)N * ((H + 15) / 16) * ((W + 15) / 16) * 4 <= kResidentWaves;
}
// K-list slots per pixel kept in LDS (the rest spills to global memory): what fits four workgroups per CU (40 KB
// each).  Round 1 traded occupancy for LDS residency on dense meshes (three / two workgroups per CU), because lists
// leaving LDS made the in-loop compactions and the lane = pixel full-list route dear; with both gone (appends beyond
// the cap are fire-and-forget stores) occupancy wins everywhere: 5120-face meshes 286 -> 250 us, 20480 faces at
// 256^2 / K=100 1.15 -> 0.85 ms.
// The fused backward's table shapes (see k_render_bwd): more face slots and fewer texel slots when the mesh is dense
// on the image (>= 0.2 faces per pixel) AND a bin sees few texels (texture no finer than half the image's resolution:
// a 16x16-pixel bin then spans at most ~8x8 texels of an object that fills the frame).  Face density alone sent 64x64
// images under 64x64 textures -- a texel per pixel, 256 texels per bin -- to the 256-slot texel table: a quarter of the
// backward's time went to collisions falling through to global atomics (pose step 192 x 64x64: 94.7 -> 71.2 us).
bool render_bwd_dense(int64_t total_faces, int N, int H, int W, int Ht, int Wt) {
  const bool dense = (double)total_faces / (double)(N > 0 ? N : 1) / ((double)H * (double)W) >= 0.2;
  return dense && (int64_t)Ht * Wt * 4 <= (int64_t)H * W;
}

// The fragment-writing build (lane = pixel, keep-the-K-smallest in place) still prefers its lists in LDS on dense
// meshes: three workgroups per CU from 0.2 faces per pixel, two when K > 64 as well.
int raster_cap(int64_t total_faces, int N, int H, int W, int K, bool fused = true, bool g8 = false) {
  const int KL = list_cap(K);
  const double density = (double)total_faces / (double)(N > 0 ? N : 1) / ((double)H * (double)W);
  // (G8: two workgroups of eight waves per CU, 80 KB each; a slot of all 256 pixels of a bin is 2 KB either way)
  const size_t budget = g8 ? (size_t)2 * kLdsBudget
                           : ((fused || density < 0.2) ? (size_t)kLdsBudget : (K > 64 ? (size_t)80 * 1024 : (size_t)52 * 1024));
  const int fit = (int)((budget - raster_fixed_lds(g8 ? 8 : 4)) / (4 * 64 * 8));
  return KL < fit ? KL : fit;
}

// largest-key words of full lists: one per pixel of every 64-pixel group (fused mode)
size_t full_thr_bytes(int N, int H, int W) { return (size_t)(N < 0 ? 0 : N) * wave_groups(H, W) * 64 * 8; }

size_t raster_spill_bytes(int64_t total_faces, int N, int H, int W, int K) {
  const int over = list_cap(K) - raster_cap(total_faces, N, H, W, K);
  return over <= 0 ? 0 : (size_t)N * wave_groups(H, W) * (size_t)over * 64 * 8;
}

template <int MODE, bool BIGK, bool MASKS, bool G8 = false> int launch_raster_k(RasterParams P, hipStream_t st) {
  P.nbx = (P.W + 15) / 16;
  P.nby = (P.H + 15) / 16;
  P.KL = list_cap(P.K);
  P.cap = (MODE == 1) ? 1 : raster_cap(P.total_faces, P.N, P.H, P.W, P.K, MODE == 2, G8);
  // the workspace's spill area is sized for the four-wave build's cap: the cap in use must not be smaller
  if (MODE != 1 && P.cap < raster_cap(P.total_faces, P.N, P.H, P.W, P.K)) return fail(UR_ERR_WORKSPACE, "K-list cap");
  const size_t lds = raster_fixed_lds(G8 ? 8 : 4) + (MODE == 1 ? 0 : (size_t)4 * P.cap * 64 * 8);
  static LdsSlot lds_slot;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&k_raster<MODE, BIGK, MASKS, G8>), lds, lds_slot)) return rc;
  const int64_t blocks = (int64_t)P.N * P.nbx * P.nby;
  if (blocks > (int64_t)kBinIndex) return fail(UR_ERR_INVALID_ARG, "grid too large");  // bit 30 of an entry = kEmptyBin
  {
    ProfScope prof(MODE == 2 ? UR_PROFILE_RASTER_FWD : UR_PROFILE_OTHER, st);
    hipLaunchKernelGGL((k_raster<MODE, BIGK, MASKS, G8>), dim3((unsigned)blocks), dim3(G8 ? 512 : 256), lds, st, P);
  }
  return check_launch("k_raster");
}

template <int MODE> int launch_raster(RasterParams P, hipStream_t st) {
  if constexpr (MODE == 2) {
    const bool big = list_cap(P.K) > 64;
    if (fused_g8(P.N, P.H, P.W, P.K)) {
      if (P.masks) return big ? launch_raster_k<2, true, true, true>(P, st) : launch_raster_k<2, false, true, true>(P, st);
      return big ? launch_raster_k<2, true, false, true>(P, st) : launch_raster_k<2, false, false, true>(P, st);
    }
    if (P.masks) return big ? launch_raster_k<2, true, true>(P, st) : launch_raster_k<2, false, true>(P, st);
  }
  P.masks = nullptr;
  if (MODE != 1 && list_cap(P.K) > 64) return launch_raster_k<MODE, true, false>(P, st);
  return launch_raster_k<MODE, false, false>(P, st);
}

// workspace = [bbox: NF float4][records: NF x 5 float4][nearest depth: NF float, padded to 16 B]
//             [schedule: N * bins int, padded to 16 B][bin loads: N * bins int, padded, x prep_parts]
//             [face masks: N * bins * prep_parts x 8 words, where the per-mesh prep kernel can make them]
//             [largest keys of full K-lists: 64 x u64 per 64-pixel group][K-list spill]
size_t sched_ints(int N, int H, int W) { return (((size_t)N * ((H + 15) / 16) * ((W + 15) / 16)) + 3) / 4 * 4; }

// equal-sized meshes of F faces: can the per-mesh prep kernel serve them, and can it leave face masks?
bool mesh_prep_ok(int64_t F, int H, int W) {
  return F > 0 && F <= kMeshPrepMaxFaces && (int64_t)((H + 15) / 16) * ((W + 15) / 16) <= kPrepBins;
}
bool mesh_masks_ok(int64_t F, int H, int W) {
  const int nbx = (W + 15) / 16, nby = (H + 15) / 16;
  return mesh_prep_ok(F, H, W) && nbx * nby <= kMaskBins && nbx <= kMaskAxis && nby <= kMaskAxis;
}
size_t mask_words(int64_t NF, int N, int H, int W) {
  const int64_t F = N > 0 ? (NF + N - 1) / N : 0;
  if (!mesh_masks_ok(F, H, W)) return 0;
  return (size_t)N * ((H + 15) / 16) * ((W + 15) / 16) * (size_t)prep_parts(NF, N) * 8;
}

const int* sched_order(const void* ws, int64_t NF, int N, int H, int W) {
  const float* fz = reinterpret_cast<const float*>(reinterpret_cast<const float4*>(ws) + 6 * NF);
  return reinterpret_cast<const int*>(fz + ((NF + 3) / 4) * 4);
}

// lays the workspace out and points P at its parts
void carve_workspace(void* ws, int64_t NF, int N, int H, int W, RasterParams& P, int*& binload, int*& order,
                     unsigned** masks = nullptr) {
  float4* bbox = reinterpret_cast<float4*>(ws);
  float4* rec = bbox + NF;
  float* fz = reinterpret_cast<float*>(rec + 5 * NF);
  order = reinterpret_cast<int*>(fz + ((NF + 3) / 4) * 4);
  binload = order + sched_ints(N, H, W);
  P.bbox = bbox; P.rec = rec; P.fz = fz; P.order = order; P.total_faces = NF;
  unsigned* mk = reinterpret_cast<unsigned*>(binload + sched_ints(N, H, W) * prep_parts(NF, N));
  if (masks) *masks = mk;
  P.full_thr = reinterpret_cast<unsigned long long*>(mk + mask_words(NF, N, H, W));
  P.spill = P.full_thr + full_thr_bytes(N, H, W) / 8;
}

// Equal-sized meshes: does the one-call forward go through the per-mesh prep kernel?
bool one_call_prep(int N, int V, int F, int H, int W, int Ht, int Wt) {
  const int64_t nbins = (int64_t)((H + 15) / 16) * ((W + 15) / 16);
  return mesh_prep_ok(F, H, W) && (int64_t)F * 9 <= 0x7fffffffLL && (int64_t)Ht * Wt * 3 <= 0x7fffffffLL &&
         N <= kBinIndex / nbins && N <= 65535;
}

int prep_faces(const float* face_verts, int64_t NF, float blur, int cull, const int64_t* first, int N, int H, int W,
               int K, void* ws, size_t ws_bytes, hipStream_t st, RasterParams& P) {
  const size_t need = ur_workspace_bytes(NF, N, H, W, K);
  if (ws_bytes < need) return fail(UR_ERR_WORKSPACE, "workspace too small: %zu < %zu", ws_bytes, need);
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 15)) return fail(UR_ERR_WORKSPACE, "workspace must be 16-byte aligned");
  int* binload = nullptr;
  int* order = nullptr;
  carve_workspace(ws, NF, N, H, W, P, binload, order);
  float4* bbox = const_cast<float4*>(P.bbox);
  float4* rec = const_cast<float4*>(P.rec);
  float* fz = const_cast<float*>(P.fz);
  const int64_t n_bins64 = (int64_t)N * ((H + 15) / 16) * ((W + 15) / 16);
  if (n_bins64 > (int64_t)kBinIndex) return fail(UR_ERR_INVALID_ARG, "grid too large");
  const int n_bins = (int)n_bins64;
  if (hipMemsetAsync(binload, 0, (size_t)n_bins * 4, st) != hipSuccess) return fail(UR_ERR_LAUNCH, "hipMemsetAsync");
  if (NF > 0) {
    hipLaunchKernelGGL(k_face_prep, dim3((unsigned)((NF + 255) / 256)), dim3(256), 0, st, face_verts, NF,
                       sqrtf(blur), cull, bbox, rec, fz, first, N, H, W, binload);
    int rc = check_launch("k_face_prep");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_bin_order, dim3(1), dim3(1024), 0, st, binload, n_bins, order, 1, n_bins, n_bins);
  return check_launch("k_bin_order");
}

// ------------------------------------------------------------------------------------------------------------------
// rasterizer backward: one thread per (pixel, k) entry
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void edge_bwd(float px, float py, float ax, float ay, float bx, float by, float g,
                                         float& gax, float& gay, float& gbx, float& gby) {
  gax += g * (py - by);
  gay += g * (bx - px);
  gbx += -g * (py - ay);
  gby += g * (px - ax);
}

__device__ __forceinline__ void seg_bwd(float px, float py, float ax, float ay, float bx, float by, float g, float& gax,
                                        float& gay, float& gbx, float& gby) {
  const float bax = bx - ax, bay = by - ay, pax = px - ax, pay = py - ay;
  const float l2 = bax * bax + bay * bay;
  if (l2 <= kEps) {
    gbx += -(2.0f * (px - bx) * g);
    gby += -(2.0f * (py - by) * g);
    return;
  }
  const float t = (bax * pax + bay * pay) / l2;
  const float tt = min_s(max_s(t, 0.0f), 1.0f);
  const float qx = ax + tt * bax, qy = ay + tt * bay;
  const float basex = g * 2.0f * (px - qx), basey = g * 2.0f * (py - qy);
  gax += basex * (tt - 1.0f);
  gay += basey * (tt - 1.0f);
  gbx += -(basex * tt);
  gby += -(basey * tt);
}

// gradient of one fragment wrt its face's nine coordinates, given upstream (gz, gb[3], gd)
__device__ __forceinline__ void fragment_bwd(const float* __restrict__ fv, float px, float py, bool clip, float gz,
                                             float gb0, float gb1, float gb2, float gdist, float out[9]) {
  const float v0x = fv[0], v0y = fv[1], z0 = fv[2], v1x = fv[3], v1y = fv[4], z1 = fv[5], v2x = fv[6], v2y = fv[7],
              z2 = fv[8];
  const float denom = ((v2x - v0x) * (v1y - v0y) - (v2y - v0y) * (v1x - v0x)) + kEps;
  const float e0 = (px - v1x) * (v2y - v1y) - (py - v1y) * (v2x - v1x);
  const float e1 = (px - v2x) * (v0y - v2y) - (py - v2y) * (v0x - v2x);
  const float e2 = (px - v0x) * (v1y - v0y) - (py - v0y) * (v1x - v0x);
  const float w0 = e0 / denom, w1 = e1 / denom, w2 = e2 / denom;
  float c0 = w0, c1 = w1, c2 = w2, s = 1.0f;
  bool sum_clamped = false;
  if (clip) {
    c0 = max_s(w0, 0.0f); c1 = max_s(w1, 0.0f); c2 = max_s(w2, 0.0f);
    const float sraw = c0 + c1 + c2;
    sum_clamped = sraw < 1e-5f;
    s = max_s(sraw, 1e-5f);
  }
  const float g0 = gb0 + gz * z0, g1 = gb1 + gz * z1, g2 = gb2 + gz * z2;
  float gw0, gw1, gw2;
  if (clip) {
    const float gs = sum_clamped ? 0.0f : -(g0 * c0 + g1 * c1 + g2 * c2) / (s * s);
    gw0 = (w0 > 0.0f) ? (g0 / s + gs) : 0.0f;
    gw1 = (w1 > 0.0f) ? (g1 / s + gs) : 0.0f;
    gw2 = (w2 > 0.0f) ? (g2 / s + gs) : 0.0f;
  } else {
    gw0 = g0; gw1 = g1; gw2 = g2;
  }
  const float ge0 = gw0 / denom, ge1 = gw1 / denom, ge2 = gw2 / denom;
  const float gden = -(gw0 * w0 + gw1 * w1 + gw2 * w2) / denom;
  float a0x = 0, a0y = 0, a1x = 0, a1y = 0, a2x = 0, a2y = 0;
  edge_bwd(px, py, v1x, v1y, v2x, v2y, ge0, a1x, a1y, a2x, a2y);
  edge_bwd(px, py, v2x, v2y, v0x, v0y, ge1, a2x, a2y, a0x, a0y);
  edge_bwd(px, py, v0x, v0y, v1x, v1y, ge2, a0x, a0y, a1x, a1y);
  // denom = edge(v2; v0, v1) + eps : v2 plays the role of the point
  edge_bwd(v2x, v2y, v0x, v0y, v1x, v1y, gden, a0x, a0y, a1x, a1y);
  a2x += gden * (v1y - v0y);
  a2y += -gden * (v1x - v0x);
  const bool inside = (w0 > 0.0f) && (w1 > 0.0f) && (w2 > 0.0f);
  const float gd = inside ? -gdist : gdist;
  const float l01 = (v1x - v0x) * (v1x - v0x) + (v1y - v0y) * (v1y - v0y);
  const float l02 = (v2x - v0x) * (v2x - v0x) + (v2y - v0y) * (v2y - v0y);
  const float l12 = (v2x - v1x) * (v2x - v1x) + (v2y - v1y) * (v2y - v1y);
  const float d01 = seg_dist(px, py, v0x, v0y, v1x, v1y, v1x - v0x, v1y - v0y, l01, 1.0f / l01);
  const float d02 = seg_dist(px, py, v0x, v0y, v2x, v2y, v2x - v0x, v2y - v0y, l02, 1.0f / l02);
  const float d12 = seg_dist(px, py, v1x, v1y, v2x, v2y, v2x - v1x, v2y - v1y, l12, 1.0f / l12);
  if (d01 <= d02 && d01 <= d12) seg_bwd(px, py, v0x, v0y, v1x, v1y, gd, a0x, a0y, a1x, a1y);
  else if (d02 <= d01 && d02 <= d12) seg_bwd(px, py, v0x, v0y, v2x, v2y, gd, a0x, a0y, a2x, a2y);
  else seg_bwd(px, py, v1x, v1y, v2x, v2y, gd, a1x, a1y, a2x, a2y);
  out[0] = a0x; out[1] = a0y; out[2] = gz * (c0 / s);
  out[3] = a1x; out[4] = a1y; out[5] = gz * (c1 / s);
  out[6] = a2x; out[7] = a2y; out[8] = gz * (c2 / s);
}

__global__ void __launch_bounds__(256) k_raster_bwd(const float* __restrict__ fv, const int32_t* __restrict__ p2f,
                                                    const float* __restrict__ gz, const float* __restrict__ gb,
                                                    const float* __restrict__ gd, int N, int H, int W, int K, int clip,
                                                    float* __restrict__ gfv) {
  const int64_t total = (int64_t)N * H * W * K;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int fid = p2f[e];
    if (fid < 0) continue;
    const int64_t pix = e / K;
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    const float px = pix_to_ndc(W - 1 - x, W, H), py = pix_to_ndc(H - 1 - y, H, W);
    const float ugz = gz ? gz[e] : 0.0f, ugd = gd ? gd[e] : 0.0f;
    const float ub0 = gb ? gb[e * 3 + 0] : 0.0f, ub1 = gb ? gb[e * 3 + 1] : 0.0f, ub2 = gb ? gb[e * 3 + 2] : 0.0f;
    float g[9];
    fragment_bwd(fv + (int64_t)fid * 9, px, py, clip != 0, ugz, ub0, ub1, ub2, ugd, g);
#pragma unroll
    for (int i = 0; i < 9; ++i)
      if (g[i] != 0.0f) atomicAdd(gfv + (int64_t)fid * 9 + i, g[i]);
  }
}


// ------------------------------------------------------------------------------------------------------------------
// fused backward of the layered render: grad_image -> grad_face_verts (xy only: the layered blend never reads zbuf),
// grad_maps.  Same decomposition as the forward: workgroup = 16x16 bin, wave = 8x8 tile, lane = pixel.
//
// With occ_k = prod_{j<k}(1-a_j), behind_k = colour composited from the layers behind k, dotted with the RGB
// cotangent g, and trans_k = prod_{j>k}(1-a_j):
//     dL/dcolour_k = occ_k a_k g                          -> texture scatter + uv -> barycentrics -> vertex xy
//     dL/da_k      = occ_k (c_k.g - behind_k) + g_A occ_k trans_k            -> signed distance -> vertex xy
// Step (a): each pixel sweeps its list back to front using (occ, alpha, colour) the forward saved per layer -- a
// few flops per layer, no geometry -- and leaves (dL/dalpha_k, occ_k a_k) in a scratch array.  Step (b): the
// expensive part (re-evaluating the face at the pixel, bilinear footprint, chain rule to the vertices) is done per
// (pixel, layer) ITEM, the tile's items dealt out evenly over the 64 lanes, so lanes stay busy whatever the depth
// distribution over pixels is.
//
// Scatter: instead of one global float atomic per (pixel, layer, component) -- 64 lanes hitting 64 different faces /
// texture rows, the slowest shape for the memory-side atomic units (~16 G atomics/s measured) -- the workgroup first
// sums in two small LDS tables keyed by face index / texel index, claimed with an LDS compare-and-swap on a tag; a
// key whose slot is taken by another key falls back to the global atomic.  One flush per workgroup at the end.
// The tables hold DOUBLES: gfx950's LDS executes ds_add_f64 at full rate (~10 clk per wave instruction, + ~12 clk per
// extra lane on the same address) but serialises ds_add_f32 lane by lane (~4 clk per active lane, ~250 clk per wave
// instruction; tools/micro/lds_atomic*.hip) -- f64 sums are 5x faster here and more accurate.
// ------------------------------------------------------------------------------------------------------------------
// Table shapes (template arguments of k_render_bwd; 4 workgroups per CU need <= 40 KB together):
//   FS face slots (x 6 sums), home slot = face index mod FS;
//   16 x TC texel slots (x 3 sums), home slot = torus over (texel row, texel column): the texels of one contiguous
//   footprint never collide.
// Coarse meshes put few faces but long strips of texels into a bin (256 / 512: 27 KB), fine meshes the other way
// round (512 / 256: 37 KB); the host picks by the mesh's face density on the image and the texels a bin sees
// (render_bwd_dense).
constexpr int kProbes = 3;       // slots tried per key (open addressing) before a contribution goes to global memory
constexpr int kProbeStep = 73;   // odd, and far from the torus neighbours that a footprint fills next

__device__ __forceinline__ void bary_path_bwd(const Face& f, const Eval& ev, float px, float py, bool clip, float g0,
                                              float g1, float g2, float o[6]) {
  // every quotient below is the correctly rounded one (div_by with RN reciprocals): same bits as plain divisions
  const float w0 = ev.w0, w1 = ev.w1, w2 = ev.w2;
  float gw0 = g0, gw1 = g1, gw2 = g2;
  if (clip) {
    const float c0 = max_s(w0, 0.0f), c1 = max_s(w1, 0.0f), c2 = max_s(w2, 0.0f);
    const float sraw = c0 + c1 + c2;
    const float s = max_s(sraw, 1e-5f);
    const float is = 1.0f / s;
    const float gs = (sraw < 1e-5f) ? 0.0f : -(g0 * c0 + g1 * c1 + g2 * c2) / (s * s);
    gw0 = (w0 > 0.0f) ? (div_by(g0, s, is) + gs) : 0.0f;
    gw1 = (w1 > 0.0f) ? (div_by(g1, s, is) + gs) : 0.0f;
    gw2 = (w2 > 0.0f) ? (div_by(g2, s, is) + gs) : 0.0f;
  }
  const float ge0 = div_by(gw0, f.denom, f.idenom), ge1 = div_by(gw1, f.denom, f.idenom),
              ge2 = div_by(gw2, f.denom, f.idenom);
  const float gden = div_by(-(gw0 * w0 + gw1 * w1 + gw2 * w2), f.denom, f.idenom);
  float a0x = 0, a0y = 0, a1x = 0, a1y = 0, a2x = 0, a2y = 0;
  edge_bwd(px, py, f.v1x, f.v1y, f.v2x, f.v2y, ge0, a1x, a1y, a2x, a2y);
  edge_bwd(px, py, f.v2x, f.v2y, f.v0x, f.v0y, ge1, a2x, a2y, a0x, a0y);
  edge_bwd(px, py, f.v0x, f.v0y, f.v1x, f.v1y, ge2, a0x, a0y, a1x, a1y);
  edge_bwd(f.v2x, f.v2y, f.v0x, f.v0y, f.v1x, f.v1y, gden, a0x, a0y, a1x, a1y);
  a2x += gden * (f.v1y - f.v0y);
  a2y += -gden * (f.v1x - f.v0x);
  o[0] += a0x; o[1] += a0y; o[2] += a1x; o[3] += a1y; o[4] += a2x; o[5] += a2y;
}

// claim-or-match a slot for `key` in an open-addressed LDS tag table of `slots` entries (a constant at every call site:
// the modulo folds to a mask for a power of two, to a multiply-high otherwise), starting at its home slot h; returns the
// slot, or -1 when kProbes slots were all taken by other keys.  A global float atomic costs far more than a few LDS round trips (4 % of the texel adds going to global memory were 7 us of a 53 us kernel).
__device__ __forceinline__ int lds_claim(int* tags, int h, int key, int slots) {
#pragma unroll
  for (int i = 0; i < kProbes; ++i) {
    const int s = (int)((unsigned)(h + i * kProbeStep) % (unsigned)slots);
    const int old = atomicCAS(&tags[s], -1, key);
    if (old == -1 || old == key) return s;
  }
  return -1;
}

// G8: the forward ran eight waves of 32 pixels per bin (k_raster<2, ., ., true>).  W8 = false: the four waves here take
// two of its pixel groups each, one after the other (wave w: groups w and w + 4); W8 = true: eight waves, one group each
// (workgroups of 512 threads).
template <int kFaceSlots, int kTexCols, bool G8, bool W8 = false>
__global__ void __launch_bounds__(W8 ? 512 : 256) k_render_bwd(const RasterParams P) {
  static_assert(G8 || !W8, "eight waves need eight groups");
  constexpr int WPB = G8 ? 8 : 4;   // pixel groups per bin
  constexpr int GP = 256 / WPB;     // pixels per group
  constexpr int NT = W8 ? 512 : 256;
  constexpr int WAVES = NT / 64;
  constexpr int kTexSlots = 16 * kTexCols;
  __shared__ int s_ftag[kFaceSlots];
  __shared__ int s_ttag[kTexSlots];
  // six sums per face slot, seven doubles apart: with a stride of six (12 banks, gcd 4 with the 64) one add instruction
  // could reach only every second bank pair
  constexpr int kFaceStride = 7;
  __shared__ double s_fval[kFaceSlots * kFaceStride];
  __shared__ double s_tval[kTexSlots * 3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = P.H, W = P.W, K = P.K;
  // same workgroup schedule and pixel ownership as the forward: heaviest bins first; wave w holds every second
  // column / row of the bin from (w & 1, w >> 1)
  const int sched = P.order[blockIdx.x];
  const int item = sched & kBinIndex;
  const int n = item / (P.nbx * P.nby);
  // where component c (corner c >> 1, x / y) of face `fid` is summed: its slot of grad_face_verts, or -- in the
  // one-call path -- the x / y sum of the vertex at that corner ([mesh][vertex][2])
  auto face_sum = [&](int fid, int c) -> float* {
    if (P.sum_V <= 0) return P.gfv + (int64_t)fid * 9 + (c >> 1) * 3 + (c & 1);
    const int64_t vid = P.faces[(int64_t)mesh_of(P, n) * P.fstride + (int64_t)(fid - n * P.eqF) * 3 + (c >> 1)];
    return P.gfv + (((int64_t)n * P.sum_V + vid) << 1) + (c & 1);
  };
  if (sched & kEmptyBin) return;  // no face box reaches the bin: the forward left no rows
  const int b = item - n * (P.nbx * P.nby);
  const int by = b / P.nbx, bx = b - by * P.nbx;
  float* gmap = P.gmaps ? P.gmaps + (int64_t)mesh_of(P, n) * P.Ht * P.Wt * 3 : nullptr;
  // this wave's pixel groups: per-pixel layer counts (left by the forward) of each.  A group's rows are in pixel-major
  // item order: the layers of a pixel are rows [pre, pre + m).
  constexpr int NG = WPB / WAVES;
  int m_g[NG], any_rows = 0;
#pragma unroll
  for (int gs = 0; gs < NG; ++gs) {
    const int grp = wave + WAVES * gs;
    const int gx0 = bx * 16 + (grp & 1), gy0 = by * 16 + (grp >> 1);
    const int gx = gx0 + 2 * (lane & 7), gy = gy0 + (G8 ? 4 : 2) * (lane >> 3);
    const bool glive = (gx0 < W) && (gy0 < H) && (gx < W) && (gy < H) && (lane < GP);
    const int64_t gid = (((int64_t)n * P.nby + by) * P.nbx + bx) * WPB + grp;
    m_g[gs] = glive ? P.layers[layer_counts_offset(P.N, P.nbx, P.nby, K) + gid * GP + lane] : 0;
    any_rows |= m_g[gs];
  }
  if (!__syncthreads_or(any_rows > 0)) return;  // nothing rendered in this bin
  for (int i = tid; i < kFaceSlots; i += NT) s_ftag[i] = -1;
  for (int i = tid; i < kFaceSlots * kFaceStride; i += NT) s_fval[i] = 0.0;
  for (int i = tid; i < kTexSlots; i += NT) s_ttag[i] = -1;
  for (int i = tid; i < kTexSlots * 3; i += NT) s_tval[i] = 0.0;
  __syncthreads();

#pragma unroll 1
  for (int gs = 0; gs < NG; ++gs) {
  const int grp = wave + WAVES * gs;
  const int tx0 = bx * 16 + (grp & 1), ty0 = by * 16 + (grp >> 1);
  const bool tile_live = (tx0 < W) && (ty0 < H);
  const int x = tx0 + 2 * (lane & 7), y = ty0 + (G8 ? 4 : 2) * (lane >> 3);
  const bool live = (x < W) && (y < H) && (lane < GP);
  const int64_t tile_id = (((int64_t)n * P.nby + by) * P.nbx + bx) * WPB + grp;
  // the group's layer rows (layout: kRowWords) and one float of scratch per row
  const int4* lay_rows = reinterpret_cast<const int4*>(P.layers + (tile_live ? tile_id : 0) * (int64_t)K * GP * kRowWords);
  float* scr_tile = P.scratch + (tile_live ? tile_id : 0) * (int64_t)K * GP;  // [item]: dL/dalpha of the layer
  const int m = (NG > 1 && gs > 0) ? m_g[NG - 1] : m_g[0];
  const int incl = wave_scan_add(m);
  const int total = __builtin_amdgcn_readlane(incl, 63);
  const int lo = 0, hi = total;
  if (hi > lo) {
    const int64_t hw = (int64_t)H * W;
    float g0 = 0, g1 = 0, g2 = 0, g3 = 0;
    if (live) {
      const float* gi = P.gimage + (int64_t)n * 4 * hw + (int64_t)y * W + x;
      g0 = gi[0]; g1 = gi[hw]; g2 = gi[2 * hw]; g3 = gi[3 * hw];
    }
    // ---- (a) dL/dalpha of every layer row, one float per row.  With f_j(x) = a_j (c_j . g) + (1 - a_j) x, the colour
    //      composited behind layer i of a pixel is f_{i+1}(f_{i+2}(... f_last(bg . g))) and the transmittance behind it
    //      the product of the (1 - a_j): a suffix composition of affine maps over the pixel's rows.  Rows are walked
    //      64 at a time from the end of the group's list (lane = row, coalesced loads, the next 64 in flight), the
    //      suffixes come from a segmented wave scan (segments = pixels) and the open segment's map is carried from
    //      batch to batch -- work in proportion to the rows, not to the deepest pixel of the group.
    {
      const float bgdot_own = g0 * P.bg[0] + g1 * P.bg[1] + g2 * P.bg[2];
      float carry_m = 1.0f, carry_c = 0.0f;  // composition of the rows after this batch that belong to the pixel of its last row
      const int first_base = ((total - 1) >> 6) << 6;
      auto load_rows = [&](int base, int4& ra, int4& rb) {
        const int i = base + 63 - lane;
        if (base >= 0 && i < total) { ra = lay_rows[(int64_t)i * 2]; rb = lay_rows[(int64_t)i * 2 + 1]; }
      };
      int4 ra_n = make_int4(0, 0, 0, 0), rb_n = ra_n;
      load_rows(first_base, ra_n, rb_n);
      for (int base = first_base; base >= 0; base -= 64) {
        const int i = base + 63 - lane;  // lane 0 takes the batch's last row: lanes run back to front
        const bool act = i < total;
        const int4 ra = ra_n, rb = rb_n;
        load_rows(base - 64, ra_n, rb_n);
        const int p = act ? ra.y : lane;
        const float q0 = __shfl(g0, p, 64), q1 = __shfl(g1, p, 64), q2 = __shfl(g2, p, 64), q3 = __shfl(g3, p, 64);
        const float bgdot = __shfl(bgdot_own, p, 64);
        const int pend = __shfl(incl, p, 64);          // one past the pixel's last row
        const float occ = __int_as_float(ra.z), al = __int_as_float(ra.w);
        const float cg = q0 * __int_as_float(rb.x) + q1 * __int_as_float(rb.y) + q2 * __int_as_float(rb.z);
        // inclusive segmented scan of the maps (M, C): x -> M x + C, lane q composing on the left of lanes < q
        int F = (act && i + 1 == pend) ? 1 : 0;        // the pixel's farthest layer starts a segment
        float M = act ? 1.0f - al : 1.0f, C = act ? al * cg : 0.0f;
#define UR_SEG_STEP(CTRL, ROWMASK)                                                                                   \
        {                                                                                                            \
          const int fs = __builtin_amdgcn_update_dpp(0, F, CTRL, ROWMASK, 0xf, false);                               \
          const float ms = __int_as_float(__builtin_amdgcn_update_dpp(0x3f800000, __float_as_int(M), CTRL, ROWMASK, 0xf, false)); \
          const float cs = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(C), CTRL, ROWMASK, 0xf, false));          \
          if (!F) { C = M * cs + C; M = M * ms; }                                                                    \
          F |= fs;                                                                                                   \
        }
        UR_SEG_STEP(0x111, 0xf) UR_SEG_STEP(0x112, 0xf) UR_SEG_STEP(0x114, 0xf) UR_SEG_STEP(0x118, 0xf)
        UR_SEG_STEP(0x142, 0xa) UR_SEG_STEP(0x143, 0xc)
#undef UR_SEG_STEP
        if (!F) { C = M * carry_c + C; M = M * carry_m; }  // still inside the segment left open by the previous batch
        // exclusive value: the lane before (the previous batch's carry for lane 0), identity at a segment start
        float Me = __shfl_up(M, 1, 64), Ce = __shfl_up(C, 1, 64);
        if (lane == 0) { Me = carry_m; Ce = carry_c; }
        if (act && i + 1 == pend) { Me = 1.0f; Ce = 0.0f; }
        carry_m = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(M), 63));
        carry_c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(C), 63));
        if (act) {
          const float behind = Me * bgdot + Ce;
          scr_tile[i] = occ * (cg - behind) + q3 * occ * Me;
        }
      }
    }
    // ---- (b) the group's (pixel, layer) items dealt out evenly over the lanes: geometry is re-evaluated once per
    //      item, gradients go to the LDS tables.  Lane L takes the items [L nb, (L+1) nb) of the item list, nb =
    //      batches: in any one batch the 64 lanes then sit in 64 different neighbourhoods of the group, so fewer of
    //      them meet on the same face / texel -- an LDS atomic costs ~12 clocks per extra lane on its address.  A row
    //      names its pixel, so an item is one 16-byte load away; the next item's row is fetched a batch ahead.
    wave_lds_handoff();  // step (a)'s float per row was written by another lane of this wave (same tile, global memory)
    const int nbatch = (hi - lo + 63) >> 6;
    const int item0 = lo + lane * nbatch;
    const float own_px = pix_to_ndc(W - 1 - x, W, H), own_py = pix_to_ndc(H - 1 - y, H, W);
    const float* map = P.maps + (int64_t)mesh_of(P, n) * P.Ht * P.Wt * 3;
    const bool clip = P.clip != 0;
    int4 row_next = make_int4(0, 0, 0, 0);
    float ga_next = 0.0f;
    if (item0 < hi) { row_next = lay_rows[(int64_t)item0 * 2]; ga_next = scr_tile[item0]; }
    for (int bi = 0; bi < nbatch; ++bi) {
      const int item = item0 + bi;
      const bool act = item < hi;
      const int4 row = row_next;
      const float ga = ga_next;
      if (bi + 1 < nbatch && item + 1 < hi) { row_next = lay_rows[(int64_t)(item + 1) * 2]; ga_next = scr_tile[item + 1]; }
      const int l = act ? row.y : lane;
      // cotangent and pixel centre of the owner pixel (all lanes take part in the shuffles)
      const float q0 = __shfl(g0, l, 64), q1 = __shfl(g1, l, 64), q2 = __shfl(g2, l, 64);
      const float px = __shfl(own_px, l, 64), py = __shfl(own_py, l, 64);
      if (!act) continue;
      const int fid = row.x;
      const float a = __int_as_float(row.w);
      const float wgt = __int_as_float(row.z) * a;  // transmittance in front of the layer x its alpha
      const Face fc = load_face(P.rec, fid);
      const float* fup = face_uv_row(P, n, fid);
      const float fu[6] = {fup[0], fup[1], fup[2], fup[3], fup[4], fup[5]};  // in flight with the record
      Eval ev;
      eval_face(fc, px, py, P.blur, clip, ev);
      const float gd = ga * layer_dalpha(ev.sd, a, true, P.sigma, P.mode);
      float o[6] = {0, 0, 0, 0, 0, 0};
      if (wgt != 0.0f) {
        const float u = (ev.c0 * fu[0] + ev.c1 * fu[2]) + ev.c2 * fu[4];
        const float v = (ev.c0 * fu[1] + ev.c1 * fu[3]) + ev.c2 * fu[5];
        const Bilin bl = bilin_setup(u, v, P.Ht, P.Wt);
        int onw, one, osw, ose;
        bilin_offsets(bl, P.Ht, P.Wt, onw, one, osw, ose);
        float gc[3] = {wgt * q0, wgt * q1, wgt * q2};
        if (P.use_amb) { gc[0] *= P.amb[0]; gc[1] *= P.amb[1]; gc[2] *= P.amb[2]; }
        if (P.gfv) {
          float gix = 0.0f, giy = 0.0f;
          const float y1 = (float)(bl.y0 + 1), x1 = (float)(bl.x0 + 1), x0 = (float)bl.x0, y0 = (float)bl.y0;
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) {
            const float t0 = map[onw + ch], t1 = one >= 0 ? map[one + ch] : 0.0f;
            const float t2 = osw >= 0 ? map[osw + ch] : 0.0f, t3 = ose >= 0 ? map[ose + ch] : 0.0f;
            const float g = gc[ch];
            gix -= t0 * (y1 - bl.iy) * g; giy -= t0 * (x1 - bl.ix) * g;
            gix += t1 * (y1 - bl.iy) * g; giy -= t1 * (bl.ix - x0) * g;
            gix -= t2 * (bl.iy - y0) * g; giy += t2 * (x1 - bl.ix) * g;
            gix += t3 * (bl.iy - y0) * g; giy += t3 * (bl.ix - x0) * g;
          }
          const float gu = gix * bl.mx * 2.0f, gv = giy * bl.my * 2.0f;
          const float gb0 = gu * fu[0] + gv * fu[1], gb1 = gu * fu[2] + gv * fu[3], gb2 = gu * fu[4] + gv * fu[5];
          bary_path_bwd(fc, ev, px, py, clip, gb0, gb1, gb2, o);
        }
        if (gmap) {
          const int offs[4] = {onw, one, osw, ose};
          const float wts[4] = {bl.wnw, bl.wne, bl.wsw, bl.wse};
          // texel (row, column) straight from the footprint: flipped rows Ht-1-y0 / Ht-2-y0, columns x0 / x0+1
          const int r0 = P.Ht - 1 - bl.y0, c0 = bl.x0;
          int hs[4], tx[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int trow = r0 - (q >> 1), tcol = c0 + (q & 1);
            tx[q] = trow * P.Wt + tcol;
            hs[q] = ((trow & 15) * kTexCols) | (tcol & (kTexCols - 1));
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) hs[q] = (offs[q] >= 0) ? lds_claim(s_ttag, hs[q], tx[q], kTexSlots) : -1;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (offs[q] < 0) continue;
            if (hs[q] >= 0) {
#pragma unroll
              for (int ch = 0; ch < 3; ++ch) atomicAdd(&s_tval[hs[q] * 3 + ch], (double)(gc[ch] * wts[q]));
            } else {
#pragma unroll
              for (int ch = 0; ch < 3; ++ch) atomicAdd(gmap + offs[q] + ch, gc[ch] * wts[q]);
            }
          }
        }
      }
      if (P.gfv && gd != 0.0f) {
        // d(signed dist)/d(ends of the nearest edge); first minimal edge wins ties
        const float gs = ev.inside ? -gd : gd;
        // the edge's ends picked by selects, ONE evaluation, the results dealt to their corners by selects: the lanes
        // of a wave sit on all three edges, and three divergent copies of the segment code ran one after the other
        const bool e01 = ev.d01 <= ev.d02 && ev.d01 <= ev.d12;
        const bool e02 = !e01 && ev.d02 <= ev.d01 && ev.d02 <= ev.d12;
        const bool from0 = e01 || e02;
        const float ax = from0 ? fc.v0x : fc.v1x, ay = from0 ? fc.v0y : fc.v1y;
        const float bx = e01 ? fc.v1x : fc.v2x, by = e01 ? fc.v1y : fc.v2y;
        float gax = 0.0f, gay = 0.0f, gbx = 0.0f, gby = 0.0f;
        seg_bwd(px, py, ax, ay, bx, by, gs, gax, gay, gbx, gby);
        o[0] += from0 ? gax : 0.0f; o[1] += from0 ? gay : 0.0f;
        o[2] += from0 ? (e01 ? gbx : 0.0f) : gax; o[3] += from0 ? (e01 ? gby : 0.0f) : gay;
        o[4] += e01 ? 0.0f : gbx; o[5] += e01 ? 0.0f : gby;
      }
      if (P.gfv && (wgt != 0.0f || gd != 0.0f)) {
        const int h = lds_claim(s_ftag, (int)((unsigned)fid % (unsigned)kFaceSlots), fid, kFaceSlots);
        if (h >= 0) {
#pragma unroll
          for (int i = 0; i < 6; ++i)
            if (o[i] != 0.0f) atomicAdd(&s_fval[h * kFaceStride + i], (double)o[i]);
        } else {
#pragma unroll
          for (int i = 0; i < 6; ++i)
            if (o[i] != 0.0f) atomicAdd(face_sum(fid, i), o[i]);
        }
      }
    }
  }
  }  // pixel groups of this wave

  __syncthreads();
  if (P.gfv)
    for (int i = tid; i < kFaceSlots * 6; i += NT) {
      const int slot = i / 6, c = i - slot * 6;
      const int tag = s_ftag[slot];
      const float v = (float)s_fval[slot * kFaceStride + c];
      if (tag >= 0 && v != 0.0f) atomicAdd(face_sum(tag, c), v);
    }
  if (gmap)
    for (int i = tid; i < kTexSlots * 3; i += NT) {
      const int slot = i / 3, c = i - slot * 3;
      const int tag = s_ttag[slot];
      const float v = (float)s_tval[i];
      if (tag >= 0 && v != 0.0f) atomicAdd(gmap + (int64_t)tag * 3 + c, v);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// shading + layered blend (unfused from the rasterizer: consumes fragments).  One thread per pixel.
// ------------------------------------------------------------------------------------------------------------------
struct ShadeParams {
  const int32_t* p2f;
  const float* bary;
  const float* dists;
  const float* maps;
  const float* fuv;
  const int64_t* first;
  int fuv_shared;
  int N, H, W, K, Ht, Wt;
  float sigma;
  float bg[3];
  float amb[3];
  int use_amb;
  int mode;
  // forward
  float* image;
  // backward
  const float* gimage;
  float* gmaps;
  float* gbary;
  float* gdists;
};

__device__ __forceinline__ void frag_uv(const ShadeParams& P, int n, int fid, int64_t e, float& u, float& v,
                                        const float*& fu) {
  const int64_t fl = P.fuv_shared ? (int64_t)fid - P.first[n] : (int64_t)fid;
  fu = P.fuv + fl * 6;
  const float b0 = P.bary[e * 3 + 0], b1 = P.bary[e * 3 + 1], b2 = P.bary[e * 3 + 2];
  u = (b0 * fu[0] + b1 * fu[2]) + b2 * fu[4];
  v = (b0 * fu[1] + b1 * fu[3]) + b2 * fu[5];
}

__global__ void __launch_bounds__(256) k_shade_fwd(const ShadeParams P) {
  const int64_t npix = (int64_t)P.N * P.H * P.W;
  const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= npix) return;
  const int64_t hw = (int64_t)P.H * P.W;
  const int n = (int)(pix / hw);
  const int64_t p = pix - (int64_t)n * hw;
  const float* map = P.maps + (int64_t)n * P.Ht * P.Wt * 3;
  float occ = 1.0f, r = 0.0f, g = 0.0f, b = 0.0f;
  for (int k = 0; k < P.K; ++k) {
    const int64_t e = pix * P.K + k;
    const int fid = P.p2f[e];
    const bool valid = fid >= 0;
    if (!valid) continue;  // an empty slot: alpha 0 -- it adds +0 to the colour and leaves the transmittance as it is
    const float a = layer_alpha(P.dists[e], valid, P.sigma, P.mode);
    float u = 0.0f, v = 0.0f;
    const float* fu;
    frag_uv(P, n, fid, e, u, v, fu);
    const Bilin bl = bilin_setup(u, v, P.Ht, P.Wt);
    float cr, cg, cb;
    sample_rgb(map, bl, P.Ht, P.Wt, cr, cg, cb);
    if (P.use_amb) { cr *= P.amb[0]; cg *= P.amb[1]; cb *= P.amb[2]; }
    r += occ * a * cr; g += occ * a * cg; b += occ * a * cb;
    occ = occ * (1.0f - a);
  }
  float* img = P.image + (int64_t)n * 4 * hw + p;
  img[0] = r + occ * 1.0f * P.bg[0];
  img[hw] = g + occ * 1.0f * P.bg[1];
  img[2 * hw] = b + occ * 1.0f * P.bg[2];
  img[3 * hw] = 1.0f - occ;
}

// backward: occ_k kept in LDS ([k][thread], conflict free), then a back-to-front sweep
__global__ void __launch_bounds__(64) k_shade_bwd(const ShadeParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* s_occ = reinterpret_cast<float*>(smem);  // K * 64 floats
  const int64_t npix = (int64_t)P.N * P.H * P.W;
  const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= npix) return;
  const int t = threadIdx.x;
  const int64_t hw = (int64_t)P.H * P.W;
  const int n = (int)(pix / hw);
  const int64_t p = pix - (int64_t)n * hw;
  const float* map = P.maps + (int64_t)n * P.Ht * P.Wt * 3;
  float* gmap = P.gmaps ? P.gmaps + (int64_t)n * P.Ht * P.Wt * 3 : nullptr;
  const float* gi = P.gimage + (int64_t)n * 4 * hw + p;
  const float g0 = gi[0], g1 = gi[hw], g2 = gi[2 * hw], g3 = gi[3 * hw];
  float occ = 1.0f;
  for (int k = 0; k < P.K; ++k) {
    const int64_t e = pix * P.K + k;
    s_occ[k * 64 + t] = occ;
    occ = occ * (1.0f - layer_alpha(P.dists[e], P.p2f[e] >= 0, P.sigma, P.mode));
  }
  float behind = g0 * P.bg[0] + g1 * P.bg[1] + g2 * P.bg[2];
  float trans = 1.0f;
  for (int k = P.K - 1; k >= 0; --k) {
    const int64_t e = pix * P.K + k;
    const int fid = P.p2f[e];
    const bool valid = fid >= 0;
    if (!valid) {
      // an empty slot (most of a K-deep fragment tensor): alpha 0, no gradient anywhere, and the sweep's running
      // values pass it unchanged -- none of the texture reads and arithmetic below
      P.gdists[e] = 0.0f;
      P.gbary[e * 3 + 0] = 0.0f; P.gbary[e * 3 + 1] = 0.0f; P.gbary[e * 3 + 2] = 0.0f;
      continue;
    }
    const float d = P.dists[e];
    const float a = layer_alpha(d, valid, P.sigma, P.mode);
    const float ok = s_occ[k * 64 + t];
    float u = 0.0f, v = 0.0f;
    const float* fu = nullptr;
    frag_uv(P, n, fid, e, u, v, fu);
    const Bilin bl = bilin_setup(u, v, P.Ht, P.Wt);
    int onw, one, osw, ose;
    bilin_offsets(bl, P.Ht, P.Wt, onw, one, osw, ose);
    float tex[4][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      tex[0][c] = map[onw + c];
      tex[1][c] = one >= 0 ? map[one + c] : 0.0f;
      tex[2][c] = osw >= 0 ? map[osw + c] : 0.0f;
      tex[3][c] = ose >= 0 ? map[ose + c] : 0.0f;
    }
    float col[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float acc = 0.0f;
      acc += tex[0][c] * bl.wnw; acc += tex[1][c] * bl.wne; acc += tex[2][c] * bl.wsw; acc += tex[3][c] * bl.wse;
      col[c] = P.use_amb ? acc * P.amb[c] : acc;
    }
    const float cg = g0 * col[0] + g1 * col[1] + g2 * col[2];
    const float ga = ok * (cg - behind) + g3 * ok * trans;
    P.gdists[e] = ga * layer_dalpha(d, a, valid, P.sigma, P.mode);
    // d loss / d texel colour (before ambient): occ_k * alpha_k * g * ambient
    const float wgt = ok * a;
    float gc[3] = {wgt * g0, wgt * g1, wgt * g2};
    if (P.use_amb) { gc[0] *= P.amb[0]; gc[1] *= P.amb[1]; gc[2] *= P.amb[2]; }
    float gix = 0.0f, giy = 0.0f;
    if (wgt != 0.0f) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float g = gc[c];
        if (gmap) {
          atomicAdd(gmap + onw + c, g * bl.wnw);
          if (one >= 0) atomicAdd(gmap + one + c, g * bl.wne);
          if (osw >= 0) atomicAdd(gmap + osw + c, g * bl.wsw);
          if (ose >= 0) atomicAdd(gmap + ose + c, g * bl.wse);
        }
        const float y1 = (float)(bl.y0 + 1), x1 = (float)(bl.x0 + 1), x0 = (float)bl.x0, y0 = (float)bl.y0;
        gix -= tex[0][c] * (y1 - bl.iy) * g; giy -= tex[0][c] * (x1 - bl.ix) * g;
        gix += tex[1][c] * (y1 - bl.iy) * g; giy -= tex[1][c] * (bl.ix - x0) * g;
        gix -= tex[2][c] * (bl.iy - y0) * g; giy += tex[2][c] * (x1 - bl.ix) * g;
        gix += tex[3][c] * (bl.iy - y0) * g; giy += tex[3][c] * (bl.ix - x0) * g;
      }
    }
    float gb0 = 0.0f, gb1 = 0.0f, gb2 = 0.0f;
    if (valid) {
      const float gu = gix * bl.mx * 2.0f, gv = giy * bl.my * 2.0f;
      gb0 = gu * fu[0] + gv * fu[1]; gb1 = gu * fu[2] + gv * fu[3]; gb2 = gu * fu[4] + gv * fu[5];
    }
    P.gbary[e * 3 + 0] = gb0; P.gbary[e * 3 + 1] = gb1; P.gbary[e * 3 + 2] = gb2;
    behind = a * cg + (1.0f - a) * behind;
    trans = trans * (1.0f - a);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// camera projection fused with the face gather, and its backward
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_project_faces(const float* __restrict__ verts, const int64_t* __restrict__ faces,
                                                       int64_t fstride, const float* __restrict__ R,
                                                       const float* __restrict__ T,
                                                       float sx, float sy, int N, int V, int F,
                                                       float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (n, f, corner)
  if (i >= (int64_t)N * F * 3) return;
  const int n = (int)(i / ((int64_t)F * 3));
  const int fc = (int)(i - (int64_t)n * F * 3);
  const int64_t vi = min(max(faces[(int64_t)n * fstride + fc], (int64_t)0), (int64_t)V - 1);  // no wild reads
  float o[3];
  project_vertex(verts + ((int64_t)n * V + vi) * 3, R + (int64_t)n * 9, T + (int64_t)n * 3, sx, sy, o);
  out[i * 3 + 0] = o[0];
  out[i * 3 + 1] = o[1];
  out[i * 3 + 2] = o[2];
}

// One workgroup per mesh: per-corner gradients are first summed per VERTEX in LDS (the projection is linear in the
// upstream gradient and depends on the vertex only), then every vertex is pushed through the camera once; the R / T
// sums are reduced in the workgroup.  No global atomics; outputs are plainly stored.
__global__ void __launch_bounds__(1024) k_project_faces_bwd_mesh(const float* __restrict__ verts,
                                                                 const int64_t* __restrict__ faces, int64_t fstride,
                                                                 const float* __restrict__ R,
                                                                 const float* __restrict__ T, float sx, float sy,
                                                                 int V, int F, const float* __restrict__ gfv,
                                                                 float* __restrict__ gverts, float* __restrict__ gR,
                                                                 float* __restrict__ gT, int vwin) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // f64 sums: the LDS adds doubles at full rate (ds_add_f64) but serialises ds_add_f32 lane by lane (~4 clk/lane).
  // blockIdx.y selects a window of `vwin` vertices (meshes whose V*24 B exceed the LDS take several workgroups, each
  // scanning every corner and keeping those of its window; R / T sums then meet in zero-filled outputs by atomics).
  double* acc = reinterpret_cast<double*>(smem);  // vwin*3, then 16*12 partial sums
  float* part = reinterpret_cast<float*>(acc + (size_t)vwin * 3);
  const int n = blockIdx.x, tid = threadIdx.x;
  const int v_lo = blockIdx.y * vwin, v_hi = min(V, v_lo + vwin);
  for (int i = tid; i < vwin * 3; i += 1024) acc[i] = 0.0;
  __syncthreads();
  const int64_t* fc = faces + (int64_t)n * fstride;
  const float* g = gfv + (int64_t)n * F * 9;
  for (int c = tid; c < F * 3; c += 1024) {
    const int vi = (int)fc[c] - v_lo;
    if (vi < 0 || vi >= v_hi - v_lo) continue;
    atomicAdd(&acc[vi * 3 + 0], (double)g[c * 3 + 0]);
    atomicAdd(&acc[vi * 3 + 1], (double)g[c * 3 + 1]);
    atomicAdd(&acc[vi * 3 + 2], (double)g[c * 3 + 2]);
  }
  __syncthreads();
  const float* r = R + (int64_t)n * 9;
  const float* t = T + (int64_t)n * 3;
  float s[12];
#pragma unroll
  for (int q = 0; q < 12; ++q) s[q] = 0.0f;
  for (int v = v_lo + tid; v < v_hi; v += 1024) {
    const float* p = verts + ((int64_t)n * V + v) * 3;
    const int a = (v - v_lo) * 3;
    const float g0 = (float)acc[a + 0], g1 = (float)acc[a + 1], g2 = (float)acc[a + 2];
    const float vx = ((p[0] * r[0] + p[1] * r[3]) + p[2] * r[6]) + t[0];
    const float vy = ((p[0] * r[1] + p[1] * r[4]) + p[2] * r[7]) + t[1];
    const float vz = ((p[0] * r[2] + p[1] * r[5]) + p[2] * r[8]) + t[2];
    const float gv[3] = {g0 * sx / vz, g1 * sy / vz, g2 - (g0 * sx * vx + g1 * sy * vy) / (vz * vz)};
    if (gverts) {
      float* go = gverts + ((int64_t)n * V + v) * 3;
#pragma unroll
      for (int i = 0; i < 3; ++i) go[i] = gv[0] * r[i * 3 + 0] + gv[1] * r[i * 3 + 1] + gv[2] * r[i * 3 + 2];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) s[i * 3 + j] += p[i] * gv[j];
    s[9] += gv[0]; s[10] += gv[1]; s[11] += gv[2];
  }
  if (!gR && !gT) return;
#pragma unroll
  for (int q = 0; q < 12; ++q) {
    const float v = wave_sum(s[q]);
    if ((tid & 63) == 63) part[(tid >> 6) * 12 + q] = v;
  }
  __syncthreads();
  if (tid < 12) {
    float v = 0.0f;
    for (int w = 0; w < 16; ++w) v += part[w * 12 + tid];
    float* dst = tid < 9 ? (gR ? gR + (int64_t)n * 9 + tid : nullptr) : (gT ? gT + (int64_t)n * 3 + tid - 9 : nullptr);
    if (dst) {
      if (gridDim.y == 1) *dst = v; else atomicAdd(dst, v);
    }
  }
}

// The camera's backward from per-VERTEX sums of the NDC x / y gradients (the fused backward's one-call mode leaves
// them as [image][vertex][2]; the layered blend never reads depth, so there is no z term): the per-vertex half of
// k_project_faces_bwd_mesh, same arithmetic, nothing to gather.  One workgroup per MESH: with M meshes tiled over
// N = reps * M poses (image n = mesh n % M under pose n) it walks the mesh's poses in ascending order and adds their
// vertex gradients up (a fixed order: deterministic), so grad_verts is (M,V,3) with no copies of the meshes anywhere;
// grad_R / grad_T stay per pose.  per_pose != 0 (no grad_verts wanted -- the pose step -- or no tiling): one workgroup
// per POSE instead, nothing to add up across poses.
__global__ void __launch_bounds__(1024) k_camera_bwd_mesh(const float* __restrict__ verts, const float* __restrict__ R,
                                                          const float* __restrict__ T, float sx, float sy, int V, int M,
                                                          int reps, int per_pose, const float2* __restrict__ vsum,
                                                          float* __restrict__ gverts, float* __restrict__ gR,
                                                          float* __restrict__ gT) {
  __shared__ float part[16 * 12];
  const int tid = threadIdx.x;
  const int m = per_pose ? (int)blockIdx.x % M : (int)blockIdx.x;
  const int k_lo = per_pose ? (int)blockIdx.x / M : 0, k_hi = per_pose ? k_lo + 1 : reps;
  for (int k = k_lo; k < k_hi; ++k) {
    const int n = k * M + m;
    const float* r = R + (int64_t)n * 9;
    const float* t = T + (int64_t)n * 3;
    float s[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) s[q] = 0.0f;
    for (int v = tid; v < V; v += 1024) {
      const float* p = verts + ((int64_t)m * V + v) * 3;
      const float2 g = vsum[(int64_t)n * V + v];
      const float g0 = g.x, g1 = g.y, g2 = 0.0f;
      const float vx = ((p[0] * r[0] + p[1] * r[3]) + p[2] * r[6]) + t[0];
      const float vy = ((p[0] * r[1] + p[1] * r[4]) + p[2] * r[7]) + t[1];
      const float vz = ((p[0] * r[2] + p[1] * r[5]) + p[2] * r[8]) + t[2];
      const float gv[3] = {g0 * sx / vz, g1 * sy / vz, g2 - (g0 * sx * vx + g1 * sy * vy) / (vz * vz)};
      if (gverts) {
        // vertex v of mesh m belongs to this thread in every pose: a plain read-modify-write, poses in ascending order
        float* go = gverts + ((int64_t)m * V + v) * 3;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const float c = gv[0] * r[i * 3 + 0] + gv[1] * r[i * 3 + 1] + gv[2] * r[i * 3 + 2];
          go[i] = (k == k_lo) ? c : go[i] + c;
        }
      }
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) s[i * 3 + j] += p[i] * gv[j];
      s[9] += gv[0]; s[10] += gv[1]; s[11] += gv[2];
    }
    if (!gR && !gT) continue;
    if (k > k_lo) __syncthreads();  // the previous pose's partial sums have been read
#pragma unroll
    for (int q = 0; q < 12; ++q) {
      const float w = wave_sum(s[q]);
      if ((tid & 63) == 63) part[(tid >> 6) * 12 + q] = w;
    }
    __syncthreads();
    if (tid < 12) {
      float w = 0.0f;
      for (int wv = 0; wv < 16; ++wv) w += part[wv * 12 + tid];
      float* dst = tid < 9 ? (gR ? gR + (int64_t)n * 9 + tid : nullptr) : (gT ? gT + (int64_t)n * 3 + tid - 9 : nullptr);
      if (dst) *dst = w;
    }
  }
}

// div_by() against the hardware's IEEE division on pseudo-random operand pairs spanning the ranges the rasterizer
// produces (numerators down to 0 and to tiny cancellation residues, denominators from 1e-8 up, both signs)
__global__ void __launch_bounds__(256) k_selftest_division(unsigned long long n, unsigned seed,
                                                           unsigned long long* mismatches) {
  unsigned long long bad = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (unsigned long long)gridDim.x * blockDim.x) {
    unsigned long long x = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    const unsigned ua = (unsigned)x, ub = (unsigned)(x >> 32);
    // exponents: a in 2^[-60, 3), b in 2^[-27, 3); random mantissas and signs; every 64th numerator is 0
    float a = __uint_as_float((ua & 0x807fffffu) | ((67u + (ua >> 23) % 63u) << 23));
    const float b = __uint_as_float((ub & 0x807fffffu) | ((100u + (ub >> 23) % 30u) << 23));
    if ((i & 63) == 0) a = 0.0f;
    const float want = a / b, got = div_by(a, b, 1.0f / b);
    if (!(want == got)) ++bad;
  }
  if (bad) atomicAdd(mismatches, bad);
}

// ------------------------------------------------------------------------------------------------------------------
// blending of per-fragment colours that the caller computed (lit colours of phong / flat / gouraud shading):
// kind 0..2 = layered_rgb_blend's three alpha modes (renderer.py:175-202), kind 3 = pytorch3d's softmax_rgb_blend
// (reached through SoftPhongShaderPlus, renderer.py:205-208).  One thread per pixel, K fragments each.
// ------------------------------------------------------------------------------------------------------------------
struct BlendParamsK {
  const float* colors;   // (N,H,W,K,3)
  const int32_t* p2f;    // (N,H,W,K)
  const float* dists;
  const float* zbuf;     // softmax only
  int N, H, W, K, kind;
  float sigma, gamma, znear, zfar;
  float bg[3];
  float* image;          // (N,4,H,W)
  const float* gimage;
  float* gcolors;
  float* gdists;
  float* gzbuf;
};

constexpr float kSoftEps = 1e-10f;

// sigma == 0 (the viz settings, renderer.py:53): sigmoid(-d/0) is a step; d == 0 would be 0/0, taken as covered
__device__ __forceinline__ float soft_prob(float d, bool valid, float sigma) {
  if (!valid) return 0.0f;
  if (sigma == 0.0f) return d <= 0.0f ? 1.0f : 0.0f;
  return 1.0f / (1.0f + expf(d / sigma));
}

__global__ void __launch_bounds__(256) k_blend_fwd(const BlendParamsK P) {
  const int64_t npix = (int64_t)P.N * P.H * P.W;
  const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= npix) return;
  const int64_t hw = (int64_t)P.H * P.W;
  const int n = (int)(pix / hw);
  const int64_t p = pix - (int64_t)n * hw;
  float* img = P.image + (int64_t)n * 4 * hw + p;
  const int64_t base = pix * P.K;
  if (P.kind < 3) {
    float occ = 1.0f, r = 0.0f, g = 0.0f, b = 0.0f;
    for (int k = 0; k < P.K; ++k) {
      const int64_t e = base + k;
      const float a = layer_alpha(P.dists[e], P.p2f[e] >= 0, P.sigma, P.kind);
      r += occ * a * P.colors[e * 3 + 0]; g += occ * a * P.colors[e * 3 + 1]; b += occ * a * P.colors[e * 3 + 2];
      occ = occ * (1.0f - a);
    }
    img[0] = r + occ * 1.0f * P.bg[0];
    img[hw] = g + occ * 1.0f * P.bg[1];
    img[2 * hw] = b + occ * 1.0f * P.bg[2];
    img[3 * hw] = 1.0f - occ;
    return;
  }
  const float zden = P.zfar - P.znear;
  float alpha = 1.0f, zmax = 0.0f;
  for (int k = 0; k < P.K; ++k) {
    const int64_t e = base + k;
    const bool valid = P.p2f[e] >= 0;
    alpha *= (1.0f - soft_prob(P.dists[e], valid, P.sigma));
    const float zinv = valid ? (P.zfar - P.zbuf[e]) / zden : 0.0f;
    zmax = (k == 0) ? zinv : max_s(zmax, zinv);
  }
  zmax = max_s(zmax, kSoftEps);
  const float delta = max_s(expf((kSoftEps - zmax) / P.gamma), kSoftEps);
  float denom = delta, r = 0.0f, g = 0.0f, b = 0.0f;
  for (int k = 0; k < P.K; ++k) {
    const int64_t e = base + k;
    const bool valid = P.p2f[e] >= 0;
    const float zinv = valid ? (P.zfar - P.zbuf[e]) / zden : 0.0f;
    const float w = soft_prob(P.dists[e], valid, P.sigma) * expf((zinv - zmax) / P.gamma);
    denom += w;
    r += w * P.colors[e * 3 + 0]; g += w * P.colors[e * 3 + 1]; b += w * P.colors[e * 3 + 2];
  }
  img[0] = (r + delta * P.bg[0]) / denom;
  img[hw] = (g + delta * P.bg[1]) / denom;
  img[2 * hw] = (b + delta * P.bg[2]) / denom;
  img[3 * hw] = 1.0f - alpha;
}

// LayeredShader(debug=True): the layered blend's per-layer alphas (+ a trailing 1) and the transmittance in front of
// every layer (renderer.py:185-195), from the same layer_alpha() the blend kernels use
__global__ void __launch_bounds__(256) k_blend_intermediates(const int32_t* __restrict__ p2f,
                                                             const float* __restrict__ dists, int64_t npix, int K,
                                                             float sigma, int mode, float* __restrict__ alpha,
                                                             float* __restrict__ occ_alpha) {
  const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= npix) return;
  float occ = 1.0f;
  for (int k = 0; k < K; ++k) {
    const float a = layer_alpha(dists[pix * K + k], p2f[pix * K + k] >= 0, sigma, mode);
    alpha[pix * (K + 1) + k] = a;
    occ_alpha[pix * (K + 1) + k] = occ;
    occ = occ * (1.0f - a);
  }
  alpha[pix * (K + 1) + K] = 1.0f;
  occ_alpha[pix * (K + 1) + K] = occ;
}

__global__ void __launch_bounds__(256) k_blend_bwd(const BlendParamsK P) {
  const int64_t npix = (int64_t)P.N * P.H * P.W;
  const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= npix) return;
  const int64_t hw = (int64_t)P.H * P.W;
  const int n = (int)(pix / hw);
  const int64_t p = pix - (int64_t)n * hw;
  const float* gi = P.gimage + (int64_t)n * 4 * hw + p;
  const float g0 = gi[0], g1 = gi[hw], g2 = gi[2 * hw], g3 = gi[3 * hw];
  const int64_t base = pix * P.K;
  if (P.kind < 3) {
    // transmittance in front of every layer is recomputed front to back on the way (two sweeps, no scratch):
    // first the total, then back to front dividing out is inexact -- so sweep forward per layer instead: O(K^2)
    // would be wasteful; keep the forward products in the gradient buffers (gdists as scratch) and read them back
    float occ = 1.0f;
    for (int k = 0; k < P.K; ++k) {
      const int64_t e = base + k;
      P.gdists[e] = occ;
      occ = occ * (1.0f - layer_alpha(P.dists[e], P.p2f[e] >= 0, P.sigma, P.kind));
    }
    float behind = g0 * P.bg[0] + g1 * P.bg[1] + g2 * P.bg[2];
    float trans = 1.0f;
    for (int k = P.K - 1; k >= 0; --k) {
      const int64_t e = base + k;
      const bool valid = P.p2f[e] >= 0;
      const float d = P.dists[e];
      const float a = layer_alpha(d, valid, P.sigma, P.kind);
      const float ok = P.gdists[e];
      const float c0 = P.colors[e * 3 + 0], c1 = P.colors[e * 3 + 1], c2 = P.colors[e * 3 + 2];
      const float cg = g0 * c0 + g1 * c1 + g2 * c2;
      const float ga = ok * (cg - behind) + g3 * ok * trans;
      P.gdists[e] = ga * layer_dalpha(d, a, valid, P.sigma, P.kind);
      const float wgt = ok * a;
      P.gcolors[e * 3 + 0] = wgt * g0; P.gcolors[e * 3 + 1] = wgt * g1; P.gcolors[e * 3 + 2] = wgt * g2;
      if (P.gzbuf) P.gzbuf[e] = 0.0f;
      behind = a * cg + (1.0f - a) * behind;
      trans = trans * (1.0f - a);
    }
    return;
  }
  // ---- softmax blend: torch autograd's rules for the published formula (max -> first maximal entry, clamp(min)
  //      passes where the input is >= the bound, prod differentiated exactly also with zero factors)
  const float zden = P.zfar - P.znear;
  float zraw = 0.0f;
  int kmax = 0, zeros = 0, kzero = -1;
  float prod_nz = 1.0f;
  for (int k = 0; k < P.K; ++k) {
    const int64_t e = base + k;
    const bool valid = P.p2f[e] >= 0;
    const float pr = soft_prob(P.dists[e], valid, P.sigma);
    const float zinv = valid ? (P.zfar - P.zbuf[e]) / zden : 0.0f;
    if (k == 0 || zinv > zraw) { zraw = zinv; kmax = k; }
    const float f = 1.0f - pr;
    if (f == 0.0f) { ++zeros; kzero = k; } else prod_nz *= f;
  }
  const float zmax = max_s(zraw, kSoftEps);
  const float delta_raw = expf((kSoftEps - zmax) / P.gamma);
  const float delta = max_s(delta_raw, kSoftEps);
  float denom = delta, S0 = delta * P.bg[0], S1 = delta * P.bg[1], S2 = delta * P.bg[2];
  for (int k = 0; k < P.K; ++k) {
    const int64_t e = base + k;
    const bool valid = P.p2f[e] >= 0;
    const float zinv = valid ? (P.zfar - P.zbuf[e]) / zden : 0.0f;
    const float w = soft_prob(P.dists[e], valid, P.sigma) * expf((zinv - zmax) / P.gamma);
    denom += w;
    S0 += w * P.colors[e * 3 + 0]; S1 += w * P.colors[e * 3 + 1]; S2 += w * P.colors[e * 3 + 2];
  }
  const float rgb_dot = g0 * (S0 / denom) + g1 * (S1 / denom) + g2 * (S2 / denom);
  float g_zmax = 0.0f;
  for (int k = 0; k < P.K; ++k) {
    const int64_t e = base + k;
    const bool valid = P.p2f[e] >= 0;
    const float pr = soft_prob(P.dists[e], valid, P.sigma);
    const float zinv = valid ? (P.zfar - P.zbuf[e]) / zden : 0.0f;
    const float ex = expf((zinv - zmax) / P.gamma);
    const float w = pr * ex;
    const float c0 = P.colors[e * 3 + 0], c1 = P.colors[e * 3 + 1], c2 = P.colors[e * 3 + 2];
    P.gcolors[e * 3 + 0] = g0 * w / denom; P.gcolors[e * 3 + 1] = g1 * w / denom; P.gcolors[e * 3 + 2] = g2 * w / denom;
    const float g_w = ((g0 * c0 + g1 * c1 + g2 * c2) - rgb_dot) / denom;
    float g_prob = g_w * ex;
    const float g_zinv = g_w * w / P.gamma;
    g_zmax -= g_w * w / P.gamma;
    const float others = zeros == 0 ? prod_nz / (1.0f - pr) : ((zeros == 1 && k == kzero) ? prod_nz : 0.0f);
    g_prob += g3 * others;
    P.gdists[e] = (valid && P.sigma > 0.0f) ? g_prob * pr * (1.0f - pr) * (-1.0f / P.sigma) : 0.0f;
    if (P.gzbuf) P.gzbuf[e] = valid ? -g_zinv / zden : 0.0f;
  }
  const float g_delta = ((g0 * P.bg[0] + g1 * P.bg[1] + g2 * P.bg[2]) - rgb_dot) / denom;
  if (delta_raw >= kSoftEps) g_zmax -= g_delta * delta_raw / P.gamma;
  if (P.gzbuf && zraw >= kSoftEps && P.p2f[base + kmax] >= 0) P.gzbuf[base + kmax] += -g_zmax / zden;
}

// ------------------------------------------------------------------------------------------------------------------
// mesh regularisers of the 3D step (unicorn.py:338-343): normal consistency over edges (utils/mesh.py:92-139) and the
// uniform Laplacian (pytorch3d mesh_laplacian_smoothing(method='uniform')), value AND gradient in one launch.
// One workgroup per mesh; all meshes share one topology (edge table + vertex adjacency, built once on the host).
//   edge table row = (v0, v1, a, b): the edge and the third vertex of each of its two faces
//   normal term    = sum_e 1 - cos(n0, n1), n0 = (v1-v0) x (a-v0), n1 = -((v1-v0) x (b-v0)), cos with norms clamped at 1e-8
//   laplacian term = sum_i || mean_{j in N(i)} v_j - v_i ||
// Edge gradients are summed per vertex in an f64 LDS table when V*24 B fit (no global atomics), else float atomics.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cross3(const float a[3], const float b[3], float o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}

__global__ void __launch_bounds__(1024) k_mesh_reg(const float* __restrict__ verts, int V, const int4* __restrict__ quads,
                                                   int E, const int* __restrict__ nbr_off, const int* __restrict__ nbr_idx,
                                                   float* __restrict__ nsum, float* __restrict__ lsum,
                                                   float* __restrict__ gN, float* __restrict__ gL,
                                                   float* __restrict__ scratch, int lds_acc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* part = reinterpret_cast<float*>(smem);          // 2 x 16 wave partial sums
  double* acc = reinterpret_cast<double*>(smem + 128);   // V*3 when lds_acc
  const int n = blockIdx.x, tid = threadIdx.x;
  const float* vp = verts + (int64_t)n * V * 3;
  float* gn = gN ? gN + (int64_t)n * V * 3 : nullptr;
  float* gl = gL ? gL + (int64_t)n * V * 3 : nullptr;
  float* u = scratch + (int64_t)n * V * 3;
  if (gn) {
    if (lds_acc) for (int i = tid; i < V * 3; i += 1024) acc[i] = 0.0;
    else for (int i = tid; i < V * 3; i += 1024) gn[i] = 0.0f;
  }
  __syncthreads();
  // ---- normal consistency
  float nloss = 0.0f;
  for (int e = tid; e < E; e += 1024) {
    const int4 q = quads[e];
    float v0[3], ev[3], pa[3], pb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      v0[c] = vp[q.x * 3 + c];
      ev[c] = vp[q.y * 3 + c] - v0[c];
      pa[c] = vp[q.z * 3 + c] - v0[c];
      pb[c] = vp[q.w * 3 + c] - v0[c];
    }
    float n0[3], n1[3];
    cross3(ev, pa, n0);
    cross3(pb, ev, n1);  // = -(e x pb)
    const float l0 = sqrtf(n0[0] * n0[0] + n0[1] * n0[1] + n0[2] * n0[2]);
    const float l1 = sqrtf(n1[0] * n1[0] + n1[1] * n1[1] + n1[2] * n1[2]);
    const float m0 = max_s(l0, 1e-8f), m1 = max_s(l1, 1e-8f);
    const float dot = n0[0] * n1[0] + n0[1] * n1[1] + n0[2] * n1[2];
    const float cs = dot / (m0 * m1);
    nloss += 1.0f - cs;
    if (gn) {
      // d(1 - cos)/dn0 = -(n1/(m0 m1) - cos n0/m0^2 [if l0 > eps]), likewise n1
      float g0[3], g1[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        g0[c] = -(n1[c] / (m0 * m1) - (l0 > 1e-8f ? cs * n0[c] / (m0 * m0) : 0.0f));
        g1[c] = -(n0[c] / (m0 * m1) - (l1 > 1e-8f ? cs * n1[c] / (m1 * m1) : 0.0f));
      }
      // n0 = e x pa: d/de = pa x g0, d/dpa = g0 x e ; n1 = pb x e: d/dpb = e x g1, d/de = g1 x pb
      float ge[3], gpa[3], gpb[3], t[3];
      cross3(pa, g0, ge);
      cross3(g1, pb, t);
      cross3(g0, ev, gpa);
      cross3(ev, g1, gpb);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        ge[c] += t[c];
        const float gv0 = -(ge[c] + gpa[c] + gpb[c]);
        if (lds_acc) {
          atomicAdd(&acc[q.x * 3 + c], (double)gv0); atomicAdd(&acc[q.y * 3 + c], (double)ge[c]);
          atomicAdd(&acc[q.z * 3 + c], (double)gpa[c]); atomicAdd(&acc[q.w * 3 + c], (double)gpb[c]);
        } else {
          atomicAdd(&gn[q.x * 3 + c], gv0); atomicAdd(&gn[q.y * 3 + c], ge[c]);
          atomicAdd(&gn[q.z * 3 + c], gpa[c]); atomicAdd(&gn[q.w * 3 + c], gpb[c]);
        }
      }
    }
  }
  // ---- uniform Laplacian: pass 1, u_i = Lv_i / ||Lv_i|| parked in scratch
  float lloss = 0.0f;
  for (int i = tid; i < V; i += 1024) {
    const int b = nbr_off[i], e = nbr_off[i + 1];
    float s[3] = {0.0f, 0.0f, 0.0f};
    for (int j = b; j < e; ++j) {
      const int k = nbr_idx[j];
      s[0] += vp[k * 3 + 0]; s[1] += vp[k * 3 + 1]; s[2] += vp[k * 3 + 2];
    }
    const float inv = e > b ? 1.0f / (float)(e - b) : 0.0f;
    float lv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) lv[c] = s[c] * inv - vp[i * 3 + c];
    const float r = sqrtf(lv[0] * lv[0] + lv[1] * lv[1] + lv[2] * lv[2]);
    lloss += r;
    const float ir = r > 0.0f ? 1.0f / r : 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) u[i * 3 + c] = lv[c] * ir;
  }
  // block sums
  const float wn = wave_sum(nloss), wl = wave_sum(lloss);
  if ((tid & 63) == 63) { part[tid >> 6] = wn; part[16 + (tid >> 6)] = wl; }
  __threadfence_block();
  __syncthreads();
  if (tid == 0) {
    float a = 0.0f, b = 0.0f;
    for (int w = 0; w < 16; ++w) { a += part[w]; b += part[16 + w]; }
    nsum[n] = a;
    lsum[n] = b;
  }
  // ---- pass 2: d/dv_i sum_k ||Lv_k|| = sum_{j in N(i)} u_j / deg_j - u_i  (adjacency is symmetric)
  if (gl)
    for (int i = tid; i < V; i += 1024) {
      float g[3] = {-u[i * 3 + 0], -u[i * 3 + 1], -u[i * 3 + 2]};
      for (int j = nbr_off[i]; j < nbr_off[i + 1]; ++j) {
        const int k = nbr_idx[j];
        const float inv = 1.0f / (float)(nbr_off[k + 1] - nbr_off[k]);
        g[0] += u[k * 3 + 0] * inv; g[1] += u[k * 3 + 1] * inv; g[2] += u[k * 3 + 2] * inv;
      }
      gl[i * 3 + 0] = g[0]; gl[i * 3 + 1] = g[1]; gl[i * 3 + 2] = g[2];
    }
  if (gn && lds_acc)
    for (int i = tid; i < V * 3; i += 1024) gn[i] = (float)acc[i];
}

int fill_blend_params(BlendParamsK& P, const float* colors, const int32_t* p2f, const float* dists, const float* zbuf,
                      int N, int H, int W, int K, int kind, float sigma, float gamma, const float* bg, float znear,
                      float zfar) {
  if (!colors || !p2f || !dists || !bg) return fail(UR_ERR_INVALID_ARG, "null input pointer");
  if (N < 0 || H <= 0 || W <= 0 || K <= 0) return fail(UR_ERR_INVALID_ARG, "bad sizes");
  if (kind < 0 || kind > 3) return fail(UR_ERR_INVALID_ARG, "bad blend kind %d", kind);
  if ((kind == UR_BLEND_CLIP_INSIDE || kind == UR_BLEND_SIGMOID) && !(sigma > 0.0f))
    return fail(UR_ERR_INVALID_ARG, "sigma must be > 0 for soft blending");
  if (kind == UR_BLEND_SOFTMAX) {
    if (!(sigma >= 0.0f)) return fail(UR_ERR_INVALID_ARG, "sigma must be >= 0");
    if (!zbuf) return fail(UR_ERR_INVALID_ARG, "softmax blend needs zbuf");
    if (!(gamma > 0.0f) || !(zfar > znear)) return fail(UR_ERR_INVALID_ARG, "softmax blend needs gamma > 0, zfar > znear");
  }
  P.colors = colors; P.p2f = p2f; P.dists = dists; P.zbuf = zbuf;
  P.N = N; P.H = H; P.W = W; P.K = K; P.kind = kind;
  P.sigma = sigma; P.gamma = gamma; P.znear = znear; P.zfar = zfar;
  for (int i = 0; i < 3; ++i) P.bg[i] = bg[i];
  P.image = nullptr; P.gimage = nullptr; P.gcolors = nullptr; P.gdists = nullptr; P.gzbuf = nullptr;
  return UR_OK;
}

int fill_shade_params(ShadeParams& P, const int32_t* p2f, const float* bary, const float* dists, const float* maps,
                      const float* fuv, int fuv_shared, const int64_t* first, int N, int H, int W, int K, int Ht, int Wt,
                      float sigma, const float* bg, const float* amb, int mode) {
  if (!p2f || !bary || !dists || !maps || !fuv || !bg) return fail(UR_ERR_INVALID_ARG, "null input pointer");
  if (fuv_shared && !first) return fail(UR_ERR_INVALID_ARG, "face_uvs_shared needs mesh_first_face");
  if (N < 0 || H <= 0 || W <= 0 || K <= 0 || Ht <= 0 || Wt <= 0) return fail(UR_ERR_INVALID_ARG, "bad sizes");
  if (mode < 0 || mode > 2) return fail(UR_ERR_INVALID_ARG, "bad blend mode %d", mode);
  if (mode != UR_BLEND_HARD && !(sigma > 0.0f)) return fail(UR_ERR_INVALID_ARG, "sigma must be > 0 for soft blending");
  P.p2f = p2f; P.bary = bary; P.dists = dists; P.maps = maps; P.fuv = fuv; P.first = first; P.fuv_shared = fuv_shared;
  P.N = N; P.H = H; P.W = W; P.K = K; P.Ht = Ht; P.Wt = Wt; P.sigma = sigma; P.mode = mode;
  for (int i = 0; i < 3; ++i) { P.bg[i] = bg[i]; P.amb[i] = amb ? amb[i] : 1.0f; }
  P.use_amb = amb && !(amb[0] == 1.0f && amb[1] == 1.0f && amb[2] == 1.0f);
  P.image = nullptr; P.gimage = nullptr; P.gmaps = nullptr; P.gbary = nullptr; P.gdists = nullptr;
  return UR_OK;
}

}  // namespace

// ====================================================================================================================
// C ABI
// ====================================================================================================================
extern "C" {

const char* ur_last_error(void) { return g_err; }
int ur_abi_version(void) { return 3; }

int ur_profile_begin(int max_launches) {
  if (max_launches <= 0) return fail(UR_ERR_INVALID_ARG, "max_launches must be > 0");
  if (g_prof.on.load()) return fail(UR_ERR_INVALID_ARG, "profiling already on");
  g_prof.ev.assign((size_t)2 * max_launches, nullptr);
  g_prof.kind.assign((size_t)max_launches, UR_PROFILE_OTHER);
  for (auto& e : g_prof.ev)
    if (hipEventCreate(&e) != hipSuccess) return fail(UR_ERR_LAUNCH, "hipEventCreate failed");
  g_prof.next.store(0);
  g_prof.on.store(true, std::memory_order_release);
  return UR_OK;
}

int ur_profile_end(float* ms, int* kinds, int capacity, int* count) {
  if (!g_prof.on.load()) return fail(UR_ERR_INVALID_ARG, "profiling is not on");
  g_prof.on.store(false);
  const int n = std::min(g_prof.next.load(), (int)g_prof.kind.size());
  int out = 0;
  for (int i = 0; i < n; ++i) {
    float t = 0.0f;
    if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess ||
        hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess)
      continue;
    if (out < capacity && ms && kinds) { ms[out] = t; kinds[out] = g_prof.kind[i]; }
    ++out;
  }
  for (auto e : g_prof.ev) (void)hipEventDestroy(e);
  g_prof.ev.clear();
  g_prof.kind.clear();
  if (count) *count = out < capacity ? out : capacity;
  return UR_OK;
}

size_t ur_workspace_bytes(int64_t total_faces, int N, int H, int W, int K) {
  // bbox (16 B) + record (80 B) + nearest depth (4 B) per face, the workgroup schedule, then the spill area of the per-pixel K-lists (only when K exceeds what
  // LDS holds)
  const size_t nf = (size_t)(total_faces < 1 ? 1 : total_faces);
  return nf * 96 + ((nf + 3) / 4) * 16 +
         (size_t)(1 + prep_parts(total_faces, N)) * sched_ints(N < 0 ? 0 : N, H, W) * 4 +
         mask_words(total_faces, N < 0 ? 0 : N, H, W) * 4 + full_thr_bytes(N, H, W) + raster_spill_bytes(total_faces, N < 0 ? 0 : N, H, W, K < 1 ? 1 : K);
}

int ur_project_faces(const float* verts, const int64_t* faces, int64_t faces_mesh_stride, const float* R,
                     const float* T, float sx, float sy, int N, int V, int F, float* face_verts, void* stream) {
  if (!verts || !faces || !R || !T || !face_verts) return fail(UR_ERR_INVALID_ARG, "null pointer");
  if (N < 0 || V <= 0 || F < 0) return fail(UR_ERR_INVALID_ARG, "bad sizes");
  const int64_t total = (int64_t)N * F * 3;
  if (total == 0) return UR_OK;
  hipLaunchKernelGGL(k_project_faces, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, verts,
                     faces, faces_mesh_stride, R, T, sx, sy, N, V, F, face_verts);
  return check_launch("k_project_faces");
}

int ur_project_faces_backward(const float* verts, const int64_t* faces, int64_t faces_mesh_stride, const float* R,
                              const float* T, float sx, float sy, int N, int V, int F, const float* grad_face_verts,
                              float* grad_verts, float* grad_R, float* grad_T, void* stream) {
  if (!verts || !faces || !R || !T || !grad_face_verts) return fail(UR_ERR_INVALID_ARG, "null pointer");
  if (N < 0 || V <= 0 || F < 0) return fail(UR_ERR_INVALID_ARG, "bad sizes");
  if (N == 0 || F == 0 || (!grad_verts && !grad_R && !grad_T)) return UR_OK;
  // vertex windows: as few as make a window's f64 sums fit the LDS (one for the meshes of the reference's configs up
  // to level 4; a level-5 icosphere, 10242 vertices, takes two)
  const int max_win = (150 * 1024 - 16 * 12 * 4) / 24;
  const int nwin = (V + max_win - 1) / max_win;
  const int vwin = (V + nwin - 1) / nwin;
  const size_t lds = (size_t)vwin * 3 * 8 + 16 * 12 * 4;
  if (N > 65535 || nwin > 65535) return fail(UR_ERR_INVALID_ARG, "too many meshes / vertices per call");
  static LdsSlot lds_slot;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&k_project_faces_bwd_mesh), lds, lds_slot)) return rc;
  if (nwin > 1) {
    if (grad_R) (void)hipMemsetAsync(grad_R, 0, (size_t)N * 36, (hipStream_t)stream);
    if (grad_T) (void)hipMemsetAsync(grad_T, 0, (size_t)N * 12, (hipStream_t)stream);
  }
  hipLaunchKernelGGL(k_project_faces_bwd_mesh, dim3((unsigned)N, (unsigned)nwin), dim3(1024), lds, (hipStream_t)stream,
                     verts, faces, faces_mesh_stride, R, T, sx, sy, V, F, grad_face_verts, grad_verts, grad_R, grad_T,
                     vwin);
  return check_launch("k_project_faces_bwd_mesh");
}

int ur_rasterize_meshes(const float* face_verts, const int64_t* mesh_first_face, const int64_t* mesh_num_faces, int N,
                        int64_t total_faces, int H, int W, float blur_radius, int K, int perspective_correct,
                        int clip_barycentric_coords, int cull_backfaces, int32_t* pix_to_face, float* zbuf,
                        float* bary, float* dists, void* workspace, size_t workspace_bytes, void* stream) {
  if (!face_verts || !mesh_first_face || !mesh_num_faces || !pix_to_face)
    return fail(UR_ERR_INVALID_ARG, "null pointer");
  if (N < 0 || H <= 0 || W <= 0 || K <= 0 || total_faces < 0) return fail(UR_ERR_INVALID_ARG, "bad sizes");
  if (!(blur_radius >= 0.0f)) return fail(UR_ERR_INVALID_ARG, "blur_radius must be >= 0");
  if (perspective_correct) return fail(UR_ERR_UNSUPPORTED, "perspective_correct=True is not supported");
  if (total_faces > 0x7fffffffLL) return fail(UR_ERR_INVALID_ARG, "more than 2^31-1 faces");
  if (N == 0) return UR_OK;
  hipStream_t st = (hipStream_t)stream;
  RasterParams P{};
  int rc = prep_faces(face_verts, total_faces, blur_radius, cull_backfaces, mesh_first_face, N, H, W, K, workspace,
                      workspace_bytes, st, P);
  if (rc) return rc;
  P.first = mesh_first_face; P.nfaces = mesh_num_faces;
  P.N = N; P.H = H; P.W = W; P.K = K; P.blur = blur_radius; P.clip = clip_barycentric_coords;
  P.p2f = pix_to_face; P.zbuf = zbuf; P.bary = bary; P.dists = dists;
  return launch_raster<0>(P, st);
}

int ur_rasterize_meshes_static(const float* face_verts, const int64_t* mesh_first_face, const int64_t* mesh_num_faces,
                               int N, int64_t total_faces, int H, int W, float blur_radius, int K,
                               int perspective_correct, int clip_barycentric_coords, int cull_backfaces,
                               int32_t* pix_to_face, float* zbuf, float* bary, float* dists, void* stream) {
  if (N < 0 || H <= 0 || W <= 0 || K <= 0 || total_faces < 0) return fail(UR_ERR_INVALID_ARG, "bad sizes");
  const size_t need = ur_workspace_bytes(total_faces, N, H, W, K);
  if (need > (size_t)UR_STATIC_SCRATCH_BYTES)
    return fail(UR_ERR_WORKSPACE,
                "this call needs %zu bytes of scratch but the static scratch holds %zu: use ur_rasterize_meshes with a "
                "caller-allocated workspace", need, (size_t)UR_STATIC_SCRATCH_BYTES);
  void* ptr = nullptr;
  if (int rc = static_scratch(&ptr)) return rc;
  // same kernels as ur_rasterize_meshes; the bin-load counters inside the scratch are cleared by the call itself
  // (hipMemsetAsync on `ptr`), everything else in it is written before it is read
  return ur_rasterize_meshes(face_verts, mesh_first_face, mesh_num_faces, N, total_faces, H, W, blur_radius, K,
                             perspective_correct, clip_barycentric_coords, cull_backfaces, pix_to_face, zbuf, bary,
                             dists, ptr, (size_t)UR_STATIC_SCRATCH_BYTES, stream);
}

int ur_tile_faces(const float* face_verts, const int64_t* mesh_first_face, const int64_t* mesh_num_faces, int N,
                  int64_t total_faces, int H, int W, float blur_radius, int cull_backfaces, int cap, int32_t* counts,
                  int32_t* lists, void* workspace, size_t workspace_bytes, void* stream) {
  if (!face_verts || !mesh_first_face || !mesh_num_faces || !counts || !lists)
    return fail(UR_ERR_INVALID_ARG, "null pointer");
  if (N < 0 || H <= 0 || W <= 0 || cap <= 0 || total_faces < 0) return fail(UR_ERR_INVALID_ARG, "bad sizes");
  if (N == 0) return UR_OK;
  hipStream_t st = (hipStream_t)stream;
  RasterParams P{};
  int rc = prep_faces(face_verts, total_faces, blur_radius, cull_backfaces, mesh_first_face, N, H, W, 1, workspace,
                      workspace_bytes, st, P);
  if (rc) return rc;
  P.first = mesh_first_face; P.nfaces = mesh_num_faces;
  P.N = N; P.H = H; P.W = W; P.K = 1; P.blur = blur_radius; P.clip = 0;
  P.tl_counts = counts; P.tl_lists = lists; P.tl_cap = cap;
  return launch_raster<1>(P, st);
}

int ur_rasterize_meshes_backward(const float* face_verts, const int32_t* pix_to_face, const float* grad_zbuf,
                                 const float* grad_bary, const float* grad_dists, int N, int64_t total_faces, int H,
                                 int W, int K, int perspective_correct, int clip_barycentric_coords,
                                 float* grad_face_verts, void* stream) {
  if (!face_verts || !pix_to_face || !grad_face_verts) return fail(UR_ERR_INVALID_ARG, "null pointer");
  if (N < 0 || H <= 0 || W <= 0 || K <= 0 || total_faces < 0) return fail(UR_ERR_INVALID_ARG, "bad sizes");
  if (perspective_correct) return fail(UR_ERR_UNSUPPORTED, "perspective_correct=True is not supported");
  const int64_t total = (int64_t)N * H * W * K;
  if (total == 0 || (!grad_zbuf && !grad_bary && !grad_dists)) return UR_OK;
  const int64_t blocks = (total + 255) / 256;
  hipLaunchKernelGGL(k_raster_bwd, dim3((unsigned)(blocks > 65536 ? 65536 : blocks)), dim3(256), 0,
                     (hipStream_t)stream, face_verts, pix_to_face, grad_zbuf, grad_bary, grad_dists, N, H, W, K,
                     clip_barycentric_coords, grad_face_verts);
  return check_launch("k_raster_bwd");
}

int ur_shade_blend(const int32_t* pix_to_face, const float* bary, const float* dists, const float* maps,
                   const float* face_uvs, int face_uvs_shared, const int64_t* mesh_first_face, int N, int H, int W,
                   int K, int Ht, int Wt, float sigma, const float* background, const float* ambient, int mode,
                   float* image, void* stream) {
  ShadeParams P;
  int rc = fill_shade_params(P, pix_to_face, bary, dists, maps, face_uvs, face_uvs_shared, mesh_first_face, N, H, W, K,
                             Ht, Wt, sigma, background, ambient, mode);
  if (rc) return rc;
  if (!image) return fail(UR_ERR_INVALID_ARG, "null image");
  P.image = image;
  const int64_t npix = (int64_t)N * H * W;
  if (npix == 0) return UR_OK;
  hipLaunchKernelGGL(k_shade_fwd, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P);
  return check_launch("k_shade_fwd");
}

int ur_shade_blend_backward(const float* grad_image, const int32_t* pix_to_face, const float* bary,
                            const float* dists, const float* maps, const float* face_uvs, int face_uvs_shared,
                            const int64_t* mesh_first_face, int N, int H, int W, int K, int Ht, int Wt, float sigma,
                            const float* background, const float* ambient, int mode, float* grad_maps,
                            float* grad_bary, float* grad_dists, void* stream) {
  ShadeParams P;
  int rc = fill_shade_params(P, pix_to_face, bary, dists, maps, face_uvs, face_uvs_shared, mesh_first_face, N, H, W, K,
                             Ht, Wt, sigma, background, ambient, mode);
  if (rc) return rc;
  if (!grad_image || !grad_bary || !grad_dists) return fail(UR_ERR_INVALID_ARG, "null gradient pointer");
  P.gimage = grad_image; P.gmaps = grad_maps; P.gbary = grad_bary; P.gdists = grad_dists;
  const int64_t npix = (int64_t)N * H * W;
  if (npix == 0) return UR_OK;
  const size_t lds = (size_t)K * 64 * 4;
  if (lds > 150 * 1024) return fail(UR_ERR_UNSUPPORTED, "faces_per_pixel=%d too large for the blend backward", K);
  static LdsSlot lds_slot;
  if (int rc2 = ensure_dynamic_lds(reinterpret_cast<const void*>(&k_shade_bwd), lds, lds_slot)) return rc2;
  hipLaunchKernelGGL(k_shade_bwd, dim3((unsigned)((npix + 63) / 64)), dim3(64), lds, (hipStream_t)stream, P);
  return check_launch("k_shade_bwd");
}

int ur_mesh_regularizers(const float* verts, int N, int V, const int32_t* edge_quads, int E, const int32_t* nbr_offsets,
                         const int32_t* nbr_index, float* normal_sum, float* laplacian_sum, float* grad_normal,
                         float* grad_laplacian, float* scratch, void* stream) {
  if (!verts || !edge_quads || !nbr_offsets || !nbr_index || !normal_sum || !laplacian_sum || !scratch)
    return fail(UR_ERR_INVALID_ARG, "null pointer");
  if (N < 0 || V <= 0 || E < 0) return fail(UR_ERR_INVALID_ARG, "bad sizes");
  if (reinterpret_cast<uintptr_t>(edge_quads) & 15) return fail(UR_ERR_INVALID_ARG, "edge table must be 16-byte aligned");
  if (N == 0) return UR_OK;
  const size_t lds_full = 128 + (size_t)V * 3 * 8;
  const int lds_acc = lds_full <= 150 * 1024;
  const size_t lds = lds_acc ? lds_full : 128;
  static LdsSlot lds_slot;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&k_mesh_reg), lds, lds_slot)) return rc;
  hipLaunchKernelGGL(k_mesh_reg, dim3((unsigned)N), dim3(1024), lds, (hipStream_t)stream, verts, V,
                     reinterpret_cast<const int4*>(edge_quads), E, nbr_offsets, nbr_index, normal_sum, laplacian_sum,
                     grad_normal, grad_laplacian, scratch, lds_acc);
  return check_launch("k_mesh_reg");
}

int ur_blend_colors(const float* colors, const int32_t* pix_to_face, const float* dists, const float* zbuf, int N,
                    int H, int W, int K, int kind, float sigma, float gamma, const float* background, float znear,
                    float zfar, float* image, void* stream) {
  BlendParamsK P{};
  int rc = fill_blend_params(P, colors, pix_to_face, dists, zbuf, N, H, W, K, kind, sigma, gamma, background, znear,
                             zfar);
  if (rc) return rc;
  if (!image) return fail(UR_ERR_INVALID_ARG, "null output pointer");
  const int64_t npix = (int64_t)N * H * W;
  if (npix == 0) return UR_OK;
  P.image = image;
  hipLaunchKernelGGL(k_blend_fwd, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P);
  return check_launch("k_blend_fwd");
}

int ur_blend_colors_backward(const float* grad_image, const float* colors, const int32_t* pix_to_face,
                             const float* dists, const float* zbuf, int N, int H, int W, int K, int kind, float sigma,
                             float gamma, const float* background, float znear, float zfar, float* grad_colors,
                             float* grad_dists, float* grad_zbuf, void* stream) {
  BlendParamsK P{};
  int rc = fill_blend_params(P, colors, pix_to_face, dists, zbuf, N, H, W, K, kind, sigma, gamma, background, znear,
                             zfar);
  if (rc) return rc;
  if (!grad_image || !grad_colors || !grad_dists) return fail(UR_ERR_INVALID_ARG, "null gradient pointer");
  const int64_t npix = (int64_t)N * H * W;
  if (npix == 0) return UR_OK;
  P.gimage = grad_image; P.gcolors = grad_colors; P.gdists = grad_dists; P.gzbuf = grad_zbuf;
  hipLaunchKernelGGL(k_blend_bwd, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P);
  return check_launch("k_blend_bwd");
}

int ur_blend_intermediates(const int32_t* pix_to_face, const float* dists, int N, int H, int W, int K, float sigma,
                           int mode, float* alpha, float* occ_alpha, void* stream) {
  if (!pix_to_face || !dists || !alpha || !occ_alpha) return fail(UR_ERR_INVALID_ARG, "null pointer");
  if (N < 0 || H <= 0 || W <= 0 || K <= 0) return fail(UR_ERR_INVALID_ARG, "bad sizes");
  if (mode < 0 || mode > 2) return fail(UR_ERR_INVALID_ARG, "bad blend mode %d", mode);
  if (mode != UR_BLEND_HARD && !(sigma > 0.0f)) return fail(UR_ERR_INVALID_ARG, "sigma must be > 0 for soft blending");
  const int64_t npix = (int64_t)N * H * W;
  if (npix == 0) return UR_OK;
  hipLaunchKernelGGL(k_blend_intermediates, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     pix_to_face, dists, npix, K, sigma, mode, alpha, occ_alpha);
  return check_launch("k_blend_intermediates");
}

int ur_selftest_division(uint64_t n, uint32_t seed, uint64_t* mismatches, void* stream) {
  if (!mismatches) return fail(UR_ERR_INVALID_ARG, "null pointer");
  hipLaunchKernelGGL(k_selftest_division, dim3(2048), dim3(256), 0, (hipStream_t)stream, (unsigned long long)n, seed,
                     reinterpret_cast<unsigned long long*>(mismatches));
  return check_launch("k_selftest_division");
}

size_t ur_backward_scratch_bytes(int N, int H, int W, int K) {
  return (size_t)(N < 0 ? 0 : N) * wave_groups(H, W) * (size_t)K * 64 * 4;
}

size_t ur_layers_elems(int N, int H, int W, int K) {
  return (size_t)(N < 0 ? 0 : N) * wave_groups(H, W) * ((size_t)K * 64 * kRowWords + 64);
}

static int fill_render_params(RasterParams& P, const int64_t* first, const int64_t* nfaces, int N, int H, int W,
                              float blur, int K, int clip, const float* maps, const float* fuv, int fuv_shared, int Ht,
                              int Wt, float sigma, const float* bg, const float* amb, int mode) {
  if (!first || !nfaces || !maps || !fuv || !bg) return fail(UR_ERR_INVALID_ARG, "null pointer");
  if (N < 0 || H <= 0 || W <= 0 || K <= 0 || Ht <= 0 || Wt <= 0) return fail(UR_ERR_INVALID_ARG, "bad sizes");
  if (!(blur >= 0.0f)) return fail(UR_ERR_INVALID_ARG, "blur_radius must be >= 0");
  if (mode < 0 || mode > 2) return fail(UR_ERR_INVALID_ARG, "bad blend mode %d", mode);
  if (mode != UR_BLEND_HARD && !(sigma > 0.0f)) return fail(UR_ERR_INVALID_ARG, "sigma must be > 0 for soft blending");
  P.first = first; P.nfaces = nfaces; P.N = N; P.H = H; P.W = W; P.K = K; P.blur = blur; P.clip = clip;
  P.maps = maps; P.fuv = fuv; P.fuv_shared = fuv_shared; P.Ht = Ht; P.Wt = Wt; P.sigma = sigma; P.mode = mode;
  for (int i = 0; i < 3; ++i) { P.bg[i] = bg[i]; P.amb[i] = amb ? amb[i] : 1.0f; }
  P.use_amb = amb && !(amb[0] == 1.0f && amb[1] == 1.0f && amb[2] == 1.0f);
  return UR_OK;
}

int ur_render_layered_forward(const float* face_verts, const int64_t* mesh_first_face, const int64_t* mesh_num_faces,
                              int N, int64_t total_faces, int H, int W, float blur_radius, int K,
                              int clip_barycentric_coords, int cull_backfaces, const float* maps,
                              const float* face_uvs, int face_uvs_shared, int Ht, int Wt, float sigma,
                              const float* background, const float* ambient, int mode, float* image, int32_t* layers,
                              void* workspace, size_t workspace_bytes, void* stream) {
  if (!face_verts || !image || !layers) return fail(UR_ERR_INVALID_ARG, "null pointer");
  if (total_faces < 0 || total_faces > 0x7fffffffLL) return fail(UR_ERR_INVALID_ARG, "bad total_faces");
  RasterParams P{};
  int rc = fill_render_params(P, mesh_first_face, mesh_num_faces, N, H, W, blur_radius, K, clip_barycentric_coords, maps,
                              face_uvs, face_uvs_shared, Ht, Wt, sigma, background, ambient, mode);
  if (rc) return rc;
  if (N == 0) return UR_OK;
  hipStream_t st = (hipStream_t)stream;
  rc = prep_faces(face_verts, total_faces, blur_radius, cull_backfaces, mesh_first_face, N, H, W, K, workspace,
                  workspace_bytes, st, P);
  if (rc) return rc;
  P.image = image; P.layers = layers;
  return launch_raster<2>(P, st);
}

// eqF > 0: every mesh has eqF faces and mesh n starts at face n * eqF (the one-call path: no first[n] load in the kernel)
static int layered_backward(const float* grad_image, const int64_t* mesh_first_face,
                            const int64_t* mesh_num_faces, int N, int64_t total_faces, int H, int W,
                            float blur_radius, int K, int clip_barycentric_coords, const float* maps,
                            const float* face_uvs, int face_uvs_shared, int Ht, int Wt, float sigma,
                            const float* background, const float* ambient, int mode, const int32_t* layers,
                            const void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes,
                            float* grad_face_verts, float* grad_maps, void* stream, int eqF, int sum_V = 0,
                            const int64_t* faces = nullptr, int64_t fstride = 0, int M = 0) {
  if (!grad_image || !layers || !workspace || !scratch) return fail(UR_ERR_INVALID_ARG, "null pointer");
  if (scratch_bytes < ur_backward_scratch_bytes(N, H, W, K)) return fail(UR_ERR_WORKSPACE, "scratch too small");
  if (total_faces < 0 || total_faces > 0x7fffffffLL) return fail(UR_ERR_INVALID_ARG, "bad total_faces");
  if (workspace_bytes < ur_workspace_bytes(total_faces, N, H, W, K)) return fail(UR_ERR_WORKSPACE, "workspace too small");
  RasterParams P{};
  int rc = fill_render_params(P, mesh_first_face, mesh_num_faces, N, H, W, blur_radius, K, clip_barycentric_coords, maps,
                              face_uvs, face_uvs_shared, Ht, Wt, sigma, background, ambient, mode);
  if (rc) return rc;
  if (N == 0 || (!grad_face_verts && !grad_maps)) return UR_OK;
  P.bbox = reinterpret_cast<const float4*>(workspace);
  P.rec = P.bbox + total_faces;
  P.order = sched_order(workspace, total_faces, N, H, W);
  P.layers = const_cast<int32_t*>(layers);
  P.gimage = grad_image; P.gfv = grad_face_verts; P.gmaps = grad_maps;
  P.scratch = reinterpret_cast<float*>(scratch);
  P.eqF = eqF;
  P.M = M;
  if (sum_V > 0 && grad_face_verts && faces) { P.sum_V = sum_V; P.faces = faces; P.fstride = fstride; }
  P.nbx = (W + 15) / 16;
  P.nby = (H + 15) / 16;
  const int64_t blocks = (int64_t)N * P.nbx * P.nby;
  if (blocks > (int64_t)kBinIndex) return fail(UR_ERR_INVALID_ARG, "grid too large");
  {
    ProfScope prof(UR_PROFILE_RENDER_BWD, (hipStream_t)stream);
    const bool dense = render_bwd_dense(total_faces, N, H, W, Ht, Wt), g8 = fused_g8(N, H, W, K);
#ifndef UR_BWD_W8
#define UR_BWD_W8 1  // (eight-wave bins are small launches: one wave per group gave the backward 39 -> 33 us at 32 x 64x64)
#endif
    // (tried and dropped, profiles/r03_ab_bwd_split_fence.txt: two waves per 64-pixel group, the row list cut at a pixel
    // boundary -- 512-thread workgroups, four waves per SIMD instead of five: 48 -> 54 us at 32 x 128x128, faster only
    // for 16 x 256x256 / K=100 (372 -> 340 us) and slower again at 128 of those; a device-scope release fence per
    // workgroup, which a "last workgroup runs the camera's backward" fusion would need: 48 -> 103 us)
    if (g8 && UR_BWD_W8) {
      // two eight-wave workgroups per CU have 80 KB of LDS each: both tables at their larger size (45 KB).  It is the
      // FACE table that needs it: a 16x16 bin of a 64x64 image sees more than 256 faces of a 1280-face mesh, and the
      // ones that find no slot go to global atomics (32 x 64x64: 32.6 -> 26.8 us; 1024 texel slots add nothing)
      hipLaunchKernelGGL((k_render_bwd<512, 32, true, true>), dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, P);
    } else if (dense && g8)
      hipLaunchKernelGGL((k_render_bwd<512, 16, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, P);
    else if (dense)
      hipLaunchKernelGGL((k_render_bwd<512, 16, false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, P);
    else if (g8)
      hipLaunchKernelGGL((k_render_bwd<256, 32, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, P);
    else
#ifndef UR_BWD_FSLOTS
#define UR_BWD_FSLOTS 256  // (A/B, profiles/r03_ab_bwd_tables.txt: 256 slots keep five workgroups on a CU; 384 (four) and
#endif                     //  512 (three) are slower, 192 at six waves per SIMD spills)
      hipLaunchKernelGGL((k_render_bwd<UR_BWD_FSLOTS, 32, false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, P);
  }
  return check_launch("k_render_bwd");
}

int ur_render_layered_backward(const float* grad_image, const int64_t* mesh_first_face,
                               const int64_t* mesh_num_faces, int N, int64_t total_faces, int H, int W,
                               float blur_radius, int K, int clip_barycentric_coords, const float* maps,
                               const float* face_uvs, int face_uvs_shared, int Ht, int Wt, float sigma,
                               const float* background, const float* ambient, int mode, const int32_t* layers,
                               const void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes,
                               float* grad_face_verts, float* grad_maps, void* stream) {
  return layered_backward(grad_image, mesh_first_face, mesh_num_faces, N, total_faces, H, W, blur_radius, K,
                          clip_barycentric_coords, maps, face_uvs, face_uvs_shared, Ht, Wt, sigma, background, ambient,
                          mode, layers, workspace, workspace_bytes, scratch, scratch_bytes, grad_face_verts, grad_maps,
                          stream, 0);
}


// ---- whole render step over equal-sized meshes: camera + rasterizer + shader in one call each way ------------------
int ur_render_meshes_can_tile(int V, int F, int H, int W) {
  return V > 0 && F > 0 && H > 0 && W > 0 && mesh_prep_ok(F, H, W) && (int64_t)F * 9 <= 0x7fffffffLL &&
         (int64_t)V * 2 <= (int64_t)F * 9;
}

// n_meshes M: 0 or N = one mesh per image; 0 < M < N (N a multiple of M) = M meshes tiled over the N poses
static int check_tiling(int N, int& M, int V, int F, int H, int W) {
  if (M <= 0 || M == N) { M = N; return UR_OK; }
  if (M > N || N % M != 0) return fail(UR_ERR_INVALID_ARG, "n_meshes = %d does not divide the %d poses", M, N);
  if (!ur_render_meshes_can_tile(V, F, H, W))
    return fail(UR_ERR_UNSUPPORTED, "pose tiling (n_meshes < N) needs meshes the per-mesh kernels take "
                                    "(ur_render_meshes_can_tile): repeat the meshes instead");
  return UR_OK;
}

int ur_render_meshes_forward(const float* verts, const int64_t* faces, int64_t faces_mesh_stride, const float* R,
                             const float* T, float sx, float sy, int N, int n_meshes, int V, int F,
                             const int64_t* mesh_first_face,
                             const int64_t* mesh_num_faces, int H, int W, float blur_radius, int K,
                             int clip_barycentric_coords, int cull_backfaces, const float* maps, const float* face_uvs,
                             int face_uvs_shared, int Ht, int Wt, float sigma, const float* background,
                             const float* ambient, int mode, float* face_verts, float* image, int32_t* layers,
                             float* zero_grad_face_verts, float* zero_grad_maps, void* workspace,
                             size_t workspace_bytes, void* stream) {
  if (!verts || !faces || !R || !T || !face_verts || !image || !layers) return fail(UR_ERR_INVALID_ARG, "null pointer");
  if (V <= 0 || F < 0) return fail(UR_ERR_INVALID_ARG, "bad sizes");
  const int64_t NF = (int64_t)(N < 0 ? 0 : N) * F;
  if (NF > 0x7fffffffLL) return fail(UR_ERR_INVALID_ARG, "more than 2^31-1 faces");
  RasterParams P{};
  int rc = fill_render_params(P, mesh_first_face, mesh_num_faces, N, H, W, blur_radius, K, clip_barycentric_coords, maps,
                              face_uvs, face_uvs_shared, Ht, Wt, sigma, background, ambient, mode);
  if (rc) return rc;
  if (N == 0) return UR_OK;
  int M = n_meshes;
  if (int trc = check_tiling(N, M, V, F, H, W)) return trc;
  P.eqF = F;
  P.M = M;
  hipStream_t st = (hipStream_t)stream;
  const int nbins = ((H + 15) / 16) * ((W + 15) / 16);
  // the accumulator's part the one-call backward adds into: per-vertex sums, [image][vertex][2], when they fit
  const int64_t n_gfv = ((int64_t)V * 2 <= (int64_t)F * 9) ? (int64_t)V * 2 : (int64_t)F * 9, n_gmap = (int64_t)Ht * Wt * 3;
  if (one_call_prep(N, V, F, H, W, Ht, Wt)) {
    const size_t need = ur_workspace_bytes(NF, N, H, W, K);
    if (workspace_bytes < need) return fail(UR_ERR_WORKSPACE, "workspace too small: %zu < %zu", workspace_bytes, need);
    if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 15))
      return fail(UR_ERR_WORKSPACE, "workspace must be 16-byte aligned");
    int* binload = nullptr;
    int* order = nullptr;
    unsigned* masks = nullptr;
    carve_workspace(workspace, NF, N, H, W, P, binload, order, &masks);
    const int parts = prep_parts(NF, N);
    if (!mesh_masks_ok(F, H, W)) masks = nullptr;
    P.masks = masks;
    P.parts = parts;
    hipLaunchKernelGGL(k_project_prep_mesh, dim3((unsigned)parts, (unsigned)N), dim3(256), 0, st, verts, faces, faces_mesh_stride, R, T,
                       sx, sy, N, M, V, F, sqrtf(blur_radius), cull_backfaces, H, W, face_verts, const_cast<float4*>(P.bbox),
                       const_cast<float4*>(P.rec), const_cast<float*>(P.fz), binload, masks, zero_grad_face_verts, (int)n_gfv,
                       zero_grad_maps, (int)n_gmap);
    rc = check_launch("k_project_prep_mesh");
    if (rc) return rc;
    // (the schedule is built even when every workgroup of the launch is resident from the start: heaviest-first also
    // pairs heavy bins with light ones on a CU -- workgroup i and i + 256 share one -- and taking bins in index order
    // instead cost the 32 x 64x64 forward 34 -> 43 us)
    if (parts >= kBinSumParts) {
      hipLaunchKernelGGL(k_bin_sum, dim3((unsigned)((nbins + 255) / 256), (unsigned)N), dim3(256), 0, st, binload, parts, nbins);
      rc = check_launch("k_bin_sum");
      if (rc) return rc;
      hipLaunchKernelGGL(k_bin_order, dim3(1), dim3(1024), 0, st, binload, N * nbins, order, 1, nbins, parts * nbins);
      rc = check_launch("k_bin_order");
    } else {
      hipLaunchKernelGGL(k_bin_order, dim3(1), dim3(1024), 0, st, binload, N * nbins, order, parts, nbins, nbins);
      rc = check_launch("k_bin_order");
    }
    if (rc) return rc;
  } else {
    // large meshes: the per-face kernels (a mesh spread over many workgroups, globally sorted schedule)
    if (M != N) return fail(UR_ERR_UNSUPPORTED, "pose tiling needs the per-mesh prep kernel");  // (check_tiling agrees)
    rc = ur_project_faces(verts, faces, faces_mesh_stride, R, T, sx, sy, N, V, F, face_verts, stream);
    if (rc) return rc;
    rc = prep_faces(face_verts, NF, blur_radius, cull_backfaces, mesh_first_face, N, H, W, K, workspace, workspace_bytes,
                    st, P);
    if (rc) return rc;
    if (zero_grad_face_verts && hipMemsetAsync(zero_grad_face_verts, 0, (size_t)N * n_gfv * 4, st) != hipSuccess)
      return fail(UR_ERR_LAUNCH, "hipMemsetAsync");
    if (zero_grad_maps && hipMemsetAsync(zero_grad_maps, 0, (size_t)N * n_gmap * 4, st) != hipSuccess)
      return fail(UR_ERR_LAUNCH, "hipMemsetAsync");
  }
  P.image = image; P.layers = layers;
  return launch_raster<2>(P, st);
}

int ur_render_meshes_backward(const float* grad_image, const float* verts, const int64_t* faces,
                              int64_t faces_mesh_stride, const float* R, const float* T, float sx, float sy, int N,
                              int n_meshes, int V, int F, const int64_t* mesh_first_face, const int64_t* mesh_num_faces, int H, int W,
                              float blur_radius, int K, int clip_barycentric_coords, const float* maps,
                              const float* face_uvs, int face_uvs_shared, int Ht, int Wt, float sigma,
                              const float* background, const float* ambient, int mode, const int32_t* layers,
                              const void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes,
                              float* grad_face_verts, float* grad_maps, int grads_are_zero, float* grad_verts,
                              float* grad_R, float* grad_T, void* stream) {
  if (N < 0 || F < 0 || Ht <= 0 || Wt <= 0) return fail(UR_ERR_INVALID_ARG, "bad sizes");
  int M = n_meshes;
  if (N > 0)
    if (int trc = check_tiling(N, M, V, F, H, W)) return trc;
  const int64_t NF = (int64_t)N * F;
  const bool geometry = grad_verts || grad_R || grad_T;
  if (geometry && !grad_face_verts) return fail(UR_ERR_INVALID_ARG, "geometry gradients need the grad_face_verts buffer");
  hipStream_t st = (hipStream_t)stream;
  // One-call path: the fused kernel adds into per-vertex sums (kept in the grad_face_verts buffer, which they must
  // fit), so the camera's backward is a per-vertex pass with nothing to gather (k_camera_bwd_mesh).
  const bool sums_fit = (int64_t)V * 2 <= (int64_t)F * 9;
  const bool vsum = geometry && N > 0 && F > 0 && V > 0 && verts && faces && R && T && sums_fit;
  if (N > 0) {
    // the forward clears only what the per-vertex sums take (N*V*2 floats) when they fit, else the whole accumulator
    // (1.4 MB of stores less per step at 32 meshes of 1280 faces)
    if (geometry && vsum && !grads_are_zero &&
        hipMemsetAsync(grad_face_verts, 0, (size_t)N * V * 8, st) != hipSuccess)
      return fail(UR_ERR_LAUNCH, "hipMemsetAsync");
    if (geometry && !vsum && (!grads_are_zero || sums_fit) &&
        hipMemsetAsync(grad_face_verts, 0, (size_t)NF * 36, st) != hipSuccess)
      return fail(UR_ERR_LAUNCH, "hipMemsetAsync");
    if (!grads_are_zero && grad_maps && hipMemsetAsync(grad_maps, 0, (size_t)M * Ht * Wt * 12, st) != hipSuccess)
      return fail(UR_ERR_LAUNCH, "hipMemsetAsync");
  }
  int rc = layered_backward(grad_image, mesh_first_face, mesh_num_faces, N, NF, H, W, blur_radius, K,
                            clip_barycentric_coords, maps, face_uvs, face_uvs_shared, Ht, Wt, sigma, background, ambient,
                            mode, layers, workspace, workspace_bytes, scratch, scratch_bytes,
                            geometry ? grad_face_verts : nullptr, grad_maps, stream, F, vsum ? V : 0, faces,
                            faces_mesh_stride, M);
  if (rc || !geometry) return rc;
  if (vsum) {
    if (N > 65535) return fail(UR_ERR_INVALID_ARG, "too many meshes per call");
    const int per_pose = (M == N || !grad_verts) ? 1 : 0;
    hipLaunchKernelGGL(k_camera_bwd_mesh, dim3((unsigned)(per_pose ? N : M)), dim3(1024), 0, st, verts, R, T, sx, sy, V, M,
                       N / M, per_pose, reinterpret_cast<const float2*>(grad_face_verts), grad_verts, grad_R, grad_T);
    return check_launch("k_camera_bwd_mesh");
  }
  if (M != N) return fail(UR_ERR_UNSUPPORTED, "pose tiling needs per-vertex sums (V * 2 <= F * 9)");  // (check_tiling agrees)
  return ur_project_faces_backward(verts, faces, faces_mesh_stride, R, T, sx, sy, N, V, F, grad_face_verts, grad_verts,
                                   grad_R, grad_T, stream);
}

}  // extern "C"
