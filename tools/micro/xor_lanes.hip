#include <hip/hip_runtime.h>
template <int D> __device__ __forceinline__ unsigned xor_lane(unsigned v) {
  if constexpr (D == 1) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
  else if constexpr (D == 2) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);  // quad_perm [2,3,0,1]
  else if constexpr (D == 4) {
    int r = __builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xf, 0x5, false);   // row_shl:4 -> banks 0,2
    r = __builtin_amdgcn_update_dpp(r, (int)v, 0x114, 0xf, 0xa, false);       // row_shr:4 -> banks 1,3
    return (unsigned)r;
  } else if constexpr (D == 8) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, true);  // row_ror:8
  else if constexpr (D == 16) {
    auto p = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return ((threadIdx.x >> 4) & 1) ? p[0] : p[1];
  } else {
    auto p = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return ((threadIdx.x >> 5) & 1) ? p[0] : p[1];
  }
}
__global__ void k(unsigned* out) {
  unsigned v = threadIdx.x * 3 + 1;
  out[threadIdx.x] = xor_lane<1>(v);
  out[64 + threadIdx.x] = xor_lane<2>(v);
  out[128 + threadIdx.x] = xor_lane<4>(v);
  out[192 + threadIdx.x] = xor_lane<8>(v);
  out[256 + threadIdx.x] = xor_lane<16>(v);
  out[320 + threadIdx.x] = xor_lane<32>(v);
}
int main() {
  unsigned* d; hipMalloc(&d, 384 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[384]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0; int ds[6] = {1,2,4,8,16,32};
  for (int q = 0; q < 6; ++q) for (int l = 0; l < 64; ++l) if (h[q*64+l] != (unsigned)((l ^ ds[q]) * 3 + 1)) { if (bad < 10) printf("d=%d lane %d got %u\n", ds[q], l, h[q*64+l]); ++bad; }
  printf("bad %d\n", bad); return bad != 0;
}
