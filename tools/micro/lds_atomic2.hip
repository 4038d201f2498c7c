// micro-benchmark 2: ds_add_f32 vs active lanes; CAS-loop float add; ds_add_u64; ds_add_rtn_f32
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void cas_add(float* p, float v) {
  int* ip = (int*)p;
  int old = *ip;
  while (true) {
    const int want = __float_as_int(__int_as_float(old) + v);
    const int got = atomicCAS(ip, old, want);
    if (got == old) break;
    old = got;
  }
}
template <int KIND> __global__ void __launch_bounds__(256) k(int share, int active, int iters, float* out) {
  __shared__ float s[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) s[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = (lane / share) * 2 + wave * 1024;
  float acc = 0.f;
  if (lane < active)
    for (int it = 0; it < iters; ++it) {
      const int a = (slot + it * 14) & 4094;
      if (KIND == 0) atomicAdd(&s[a], 1.0f);
      else if (KIND == 1) cas_add(&s[a], 1.0f);
      else if (KIND == 2) atomicAdd((unsigned long long*)&s[a], 1ull);
      else if (KIND == 3) acc += atomicAdd(&s[a], 1.0f);
      else if (KIND == 4) atomicAdd((double*)&s[a], 1.0);
    }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = s[0] + acc;
}
int main() {
  float* out; (void)hipMalloc(&out, 4096 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 1024, iters = 1000;
  const char* names[] = {"ds_add_f32", "cas-loop f32", "ds_add_u64", "ds_add_rtn_f32", "ds_add_f64"};
  for (int kind = 0; kind < 5; ++kind)
    for (int active : {64, 16, 4, 1})
      for (int share : {1, 4, 8}) {
        if (share > active) continue;
        auto run = [&]() {
          if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, share, active, iters, out);
          if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, share, active, iters, out);
          if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, share, active, iters, out);
          if (kind == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, share, active, iters, out);
          if (kind == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, share, active, iters, out);
        };
        run(); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); run(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double winstr_per_cu = (double)blocks / 256 * 4 * iters;
        printf("%-15s active %2d share %d : %.3f ms -> %.1f cycles/wave-op/CU\n", names[kind], active, share, ms,
               ms * 1e-3 * 2.4e9 / winstr_per_cu);
      }
  return 0;
}
