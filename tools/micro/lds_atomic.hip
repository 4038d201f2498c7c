// micro-benchmark: cost of LDS atomics (ds_add_f32 / ds_add_u32) as a function of how many lanes share an address
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND> __global__ void __launch_bounds__(256) k(int share, int stride, int iters, float* out) {
  __shared__ float s[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) s[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = (lane / share) * stride + wave * 1024;
  for (int it = 0; it < iters; ++it) {
    const int a = (slot + it * 7) & 4095;
    if (KIND == 0) atomicAdd(&s[a], 1.0f);
    else if (KIND == 1) atomicAdd((unsigned*)&s[a], 1u);
    else if (KIND == 2) s[a] += 1.0f;   // plain RMW (racy), for reference
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = s[0];
}
int main() {
  float* out; hipMalloc(&out, 4096 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 1024, iters = 2000;
  for (int kind = 0; kind < 3; ++kind)
    for (int share : {1, 2, 4, 8, 16, 64})
      for (int stride : {1, 3}) {
        auto run = [&]() {
          if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, share, stride, iters, out);
          if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, share, stride, iters, out);
          if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, share, stride, iters, out);
        };
        run(); hipDeviceSynchronize();
        hipEventRecord(e0); run(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // wave-instructions per CU: blocks/256 CUs * 4 waves * iters
        const double winstr_per_cu = (double)blocks / 256 * 4 * iters;
        printf("kind %d share %2d stride %d : %.3f ms  -> %.1f cycles/wave-instr/CU (2.4 GHz)\n", kind, share, stride, ms,
               ms * 1e-3 * 2.4e9 / winstr_per_cu);
      }
  return 0;
}
