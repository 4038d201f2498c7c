// micro-benchmark: where (XCD / SE / CU) and when the dispatcher places workgroup i of a 2048-workgroup launch whose
// workgroups have the raster kernel's footprint (256 threads, 40 KB LDS -> 4 per CU), and how a weight-sorted launch
// order turns into per-CU load.  Workgroup i spins for weight(i) microseconds-ish (weights descending, like the
// heaviest-first schedule).  Output: per workgroup xcc, se, cu, start and end clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <map>

__global__ void __launch_bounds__(256, 4) k_place(const int* __restrict__ weight, unsigned long long* __restrict__ out) {
  extern __shared__ unsigned char smem[];
  const unsigned long long t0 = wall_clock64();
  unsigned hwid, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  // busy work: weight[i] rounds of dependent VALU (4 waves all spin: the SIMD is shared with the CU's other workgroups)
  float v = (float)threadIdx.x;
  const int w = weight[blockIdx.x];
  for (int r = 0; r < w; ++r)
#pragma unroll
    for (int j = 0; j < 64; ++j) v = v * 1.0000001f + 0.5f;
  if (v == 12345.678f) smem[threadIdx.x] = 1;  // keep v alive
  __syncthreads();
  if (threadIdx.x == 0) {
    out[blockIdx.x * 4 + 0] = t0;
    out[blockIdx.x * 4 + 1] = wall_clock64();
    out[blockIdx.x * 4 + 2] = hwid;
    out[blockIdx.x * 4 + 3] = xcc;
  }
}

int main(int argc, char** argv) {
  const int n = 2048;
  std::vector<int> w(n);
  // weights like the sorted bins of the headline workload: ~1/2 of the bins empty, the rest from heavy to light
  // argv[1]: 0 = weights descending in launch order (heaviest first), 1 = the first 1024 launch slots "snaked":
  // slot j of block q = j / 256 takes sorted rank q*256 + (q odd ? 255 - j%256 : j%256), so that the CU that got the
  // heaviest workgroup of one block gets the lightest of the next
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  for (int j = 0; j < n; ++j) {
    int i = j;
    if (mode == 1 && j < 1024) { const int q = j >> 8, r = j & 255; i = q * 256 + ((q & 1) ? 255 - r : r); }
    w[j] = i < 1024 ? 60 - (i * 45) / 1024 : (i < 1200 ? 5 : 0);
  }
  int* dw; unsigned long long* dout;
  hipMalloc(&dw, n * 4); hipMalloc(&dout, n * 4 * 8);
  hipMemcpy(dw, w.data(), n * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_place), hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_place, dim3(n), dim3(256), 40960, 0, dw, dout);
  hipDeviceSynchronize();
  std::vector<unsigned long long> o(n * 4);
  hipMemcpy(o.data(), dout, n * 4 * 8, hipMemcpyDeviceToHost);
  unsigned long long tmin = ~0ull, tmax = 0;
  for (int i = 0; i < n; ++i) { tmin = std::min(tmin, o[i * 4]); tmax = std::max(tmax, o[i * 4 + 1]); }
  printf("kernel span %.1f us (100 MHz clock)\n", (tmax - tmin) / 100.0);
  printf("first 48 workgroups: i xcc se sh cu start_us\n");
  for (int i = 0; i < 48; ++i) {
    const unsigned h = (unsigned)o[i * 4 + 2];
    printf("%4d  xcc %llu se %u sh %u cu %2u  start %.2f end %.2f\n", i, o[i * 4 + 3] & 15, (h >> 13) & 7, (h >> 12) & 1,
           (h >> 8) & 15, (o[i * 4] - tmin) / 100.0, (o[i * 4 + 1] - tmin) / 100.0);
  }
  // per-CU: which launch indices landed there at t ~ 0, and the sum of weights a CU received in total
  std::map<unsigned, std::vector<int>> per_cu;
  for (int i = 0; i < n; ++i) {
    const unsigned h = (unsigned)o[i * 4 + 2];
    const unsigned key = ((unsigned)(o[i * 4 + 3] & 15) << 8) | (((h >> 13) & 7) << 5) | (((h >> 12) & 1) << 4) | ((h >> 8) & 15);
    per_cu[key].push_back(i);
  }
  printf("CUs seen: %zu\n", per_cu.size());
  int shown = 0;
  double wmin = 1e30, wmax = 0;
  for (auto& kv : per_cu) {
    double sum = 0;
    for (int i : kv.second) sum += w[i];
    wmin = std::min(wmin, sum); wmax = std::max(wmax, sum);
    if (shown++ < 6) {
      printf("cu key %04x: weight sum %.0f, workgroups:", kv.first, sum);
      for (size_t j = 0; j < kv.second.size() && j < 12; ++j) printf(" %d", kv.second[j]);
      printf("\n");
    }
  }
  printf("per-CU weight sum: min %.0f max %.0f (mean %.1f)\n", wmin, wmax,
         [&] { double s = 0; for (int x : w) s += x; return s / per_cu.size(); }());
  // end time of the last workgroup per CU
  double emin = 1e30, emax = 0;
  for (auto& kv : per_cu) {
    unsigned long long e = 0;
    for (int i : kv.second) e = std::max(e, o[i * 4 + 1]);
    emin = std::min(emin, (e - tmin) / 100.0); emax = std::max(emax, (e - tmin) / 100.0);
  }
  printf("per-CU finish time: min %.1f us max %.1f us\n", emin, emax);
  return 0;
}
