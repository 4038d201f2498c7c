// micro-benchmark: what does one wave64 VALU instruction cost a gfx950 SIMD in issue time, as a function of how many
// waves share the SIMD?  Settles the VALU-issue roof bench.py prices the kernels against (VERDICT r02, weak point 2:
// the hardware guide derives 2 clocks per wave64 v_fma_f32 from the 157.3 TFLOP/s vector peak, bench.py used 4).
//
// Every wave runs a loop of 8 x 16 = 128 INDEPENDENT instructions of one opcode (8 accumulators, inline asm so that
// the compiler can neither fuse nor drop nor reorder them), `iters` times, on every SIMD of the chip: grid =
// CUs x W workgroups of 256 threads (4 waves, one per SIMD), W = waves per SIMD.  Reported per (opcode, W):
//   clk_per_instr_simd  = shader clocks (s_memtime) a SIMD spends per wave-instruction
//                       = (mean over waves of elapsed shader clocks) / (iters x 128 x W)   (the W waves of a SIMD overlap)
//   shader_ghz          = s_memtime ticks / s_memrealtime ticks x 0.1 (the constant 100 MHz counter): the clock the chip ran at
//   ginstr_chip         = wave-instructions per second over the whole chip, by the waves' own elapsed real time
// A dependent chain (1 accumulator) gives the result latency.
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/valu_issue tools/micro/valu_issue.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP16(S) S S S S S S S S S S S S S S S S

enum Op { FMA, MUL, ADD, MIN, PK_FMA, PK_MUL, PK_ADD, CNDMASK, ADD_U32, MAD_U32, RCP, FMA_DEP, PK_FMA_DEP, N_OPS };
const char* kNames[N_OPS] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_min_f32", "v_pk_fma_f32", "v_pk_mul_f32",
                             "v_pk_add_f32", "v_cndmask_b32", "v_add_u32", "v_mad_u32_u24", "v_rcp_f32",
                             "v_fma_f32 (dependent chain)", "v_pk_fma_f32 (dependent chain)"};

typedef float float2v __attribute__((ext_vector_type(2)));

template <int OP> __global__ void __launch_bounds__(256) k_issue(int iters, float seed, unsigned long long* clocks, float* sink) {
  float a[8];
  float2v p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = seed + (float)(threadIdx.x + i);
    p[i] = float2v{a[i], a[i] + 0.5f};
  }
  const float b = 1.0000001f, c = 1e-9f;
  const float2v pb = {b, b}, pc = {c, c};
  unsigned ua = threadIdx.x, ub = 3;
  __builtin_amdgcn_s_barrier();
  const unsigned long long r0 = wall_clock64();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if constexpr (OP == FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      REP16(REP8(X))
#undef X
    } else if constexpr (OP == MUL) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      REP16(REP8(X))
#undef X
    } else if constexpr (OP == ADD) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      REP16(REP8(X))
#undef X
    } else if constexpr (OP == MIN) {
#define X(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      REP16(REP8(X))
#undef X
    } else if constexpr (OP == PK_FMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
      REP16(REP8(X))
#undef X
    } else if constexpr (OP == PK_MUL) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
      REP16(REP8(X))
#undef X
    } else if constexpr (OP == PK_ADD) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
      REP16(REP8(X))
#undef X
    } else if constexpr (OP == CNDMASK) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : );
      REP16(REP8(X))
#undef X
    } else if constexpr (OP == ADD_U32) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(ub));
      REP16(REP8(X))
#undef X
    } else if constexpr (OP == MAD_U32) {
#define X(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(ub), "v"(ua));
      REP16(REP8(X))
#undef X
    } else if constexpr (OP == RCP) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      REP16(REP8(X))
#undef X
    } else if constexpr (OP == FMA_DEP) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
      REP16(REP8(X))
#undef X
    } else {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[0]) : "v"(pb), "v"(pc));
      REP16(REP8(X))
#undef X
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long r1 = wall_clock64();
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
  if ((threadIdx.x & 63) == 0) {
    clocks[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t1 - t0;
    clocks[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0;
  }
  if (s == 12345.678f) sink[0] = s;  // keeps the accumulators alive
}

template <int OP> void launch(int blocks, int iters, unsigned long long* clocks, float* sink) {
  hipLaunchKernelGGL(k_issue<OP>, dim3(blocks), dim3(256), 0, 0, iters, 1.0f, clocks, sink);
}
typedef void (*LaunchFn)(int, int, unsigned long long*, float*);

int main(int argc, char** argv) {
  int cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
  cus = prop.multiProcessorCount;
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  unsigned long long* clocks;
  float* sink;
  const int max_w = 8;
  if (hipMalloc(&clocks, (size_t)cus * max_w * 4 * 16) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  LaunchFn fns[N_OPS] = {launch<FMA>, launch<MUL>, launch<ADD>, launch<MIN>, launch<PK_FMA>, launch<PK_MUL>,
                         launch<PK_ADD>, launch<CNDMASK>, launch<ADD_U32>, launch<MAD_U32>, launch<RCP>,
                         launch<FMA_DEP>, launch<PK_FMA_DEP>};
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_rate_khz\": %d, \"iters\": %d, \"instr_per_iter\": 128, \"results\": [\n",
         prop.gcnArchName, cus, prop.clockRate, iters);
  bool first = true;
  for (int op = 0; op < N_OPS; ++op)
    for (int w : {1, 2, 4, 8}) {
      const int blocks = cus * w;
      fns[op](blocks, iters, clocks, sink);  // warm-up (code object load, clocks up)
      hipDeviceSynchronize();
      hipEventRecord(e0);
      fns[op](blocks, iters, clocks, sink);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0.0f;
      hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> h((size_t)blocks * 8);
      hipMemcpy(h.data(), clocks, h.size() * 8, hipMemcpyDeviceToHost);
      double cyc = 0.0, real = 0.0, cyc_max = 0.0;
      for (size_t i = 0; i < h.size(); i += 2) {
        cyc += (double)h[i];
        real += (double)h[i + 1];
        cyc_max = (double)h[i] > cyc_max ? (double)h[i] : cyc_max;
      }
      cyc /= (double)(h.size() / 2);
      real /= (double)(h.size() / 2);
      const double n_wave_instr = (double)iters * 128.0;
      const double shader_ghz = cyc / real * 0.1;              // s_memrealtime ticks at 100 MHz
      const double wave_s = real / 100e6;
      const double ginstr = n_wave_instr * blocks * 4 / wave_s / 1e9;
      printf("%s  {\"op\": \"%s\", \"waves_per_simd\": %d, \"event_ms\": %.4f, \"wave_us_mean\": %.2f, "
             "\"shader_ghz\": %.3f, \"clk_per_instr_simd\": %.3f, \"clk_per_instr_simd_slowest_wave\": %.3f, "
             "\"ginstr_chip\": %.1f}",
             first ? "" : ",\n", kNames[op], w, ms, wave_s * 1e6, shader_ghz, cyc / (n_wave_instr * w),
             cyc_max / (n_wave_instr * w), ginstr);
      first = false;
    }
  printf("\n]}\n");
  return 0;
}
