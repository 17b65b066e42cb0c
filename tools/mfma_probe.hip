// mfma_probe.hip -- does the matrix pipe run BESIDE the vector pipe for fp32, and is it big enough to take the demodulator's
// linear stages (VERDICT r04 "next" #2a)?  Standalone: hipcc --offload-arch=gfx950 -O3 -o mfma_probe tools/mfma_probe.hip
//
// Every wave runs ITER iterations of a loop body made of NV independent v_fma_f32 (the demodulator's instruction class) and NM
// v_mfma_f32 (16x16x4 f32: 8 passes = 32 cycles; 32x32x2 f32: 16 passes = 64 cycles), interleaved, and stamps s_memtime around
// the loop; launched with 1, 2 and 4 waves per SIMD on every CU.  Printed: shader cycles per iteration and per wave, and what
// a SIMD retires per cycle -- VALU alone, MFMA alone, both: if "both" costs max(VALU, MFMA) the pipes overlap, if it costs the
// sum they do not.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int NV, int NM, int KIND>   // KIND 0: 16x16x4 f32, 1: 32x32x2 f32
__global__ __launch_bounds__(256) void probe(float *sink, unsigned long long *cyc, int iters, float seed) {
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; i++) v[i] = seed + (float)(threadIdx.x + i);
  v4f acc4[4];
  v16f acc16[2];
#pragma unroll
  for (int i = 0; i < 4; i++) acc4[i] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 16; j++) acc16[i][j] = 0.f;
  float a = seed * 0.5f, b = seed * 0.25f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int it = 0; it < iters; it++) {
    constexpr int STEPS = NM > 0 ? NM : 1;
    constexpr int VPER = NV / STEPS;
#pragma unroll
    for (int s = 0; s < STEPS; s++) {
      if (NM > 0) {
        if (KIND == 0) acc4[s & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[s & 3], 0, 0, 0);
        else acc16[s & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc16[s & 1], 0, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < VPER; k++) {
        const int i = (s * VPER + k) & 15;
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
      }
    }
  }
  asm volatile("s_nop 0" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 16; i++) r += v[i];
#pragma unroll
  for (int i = 0; i < 4; i++) r += acc4[i].x + acc4[i].y + acc4[i].z + acc4[i].w;
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 16; j++) r += acc16[i][j];
  if (r == 12345.678f) sink[0] = r;
  if ((threadIdx.x & 63) == 0) cyc[(size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NV, int NM, int KIND>
static void run(const char *name, int waves_per_simd) {
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int blocks = cus * waves_per_simd;       // 256 threads = 4 waves = one per SIMD
  const int iters = 4000;
  float *sink;
  unsigned long long *cyc;
  hipMalloc(&sink, 4);
  hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 4);
  hipLaunchKernelGGL((probe<NV, NM, KIND>), dim3(blocks), dim3(256), 0, 0, sink, cyc, iters, 1.0f);   // warm
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((probe<NV, NM, KIND>), dim3(blocks), dim3(256), 0, 0, sink, cyc, iters, 1.0f);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * 4);
  hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost);
  double sum = 0;
  for (auto c : h) sum += (double)c;
  const double per_iter = sum / h.size() / iters;                 // cycles a wave spends per iteration (100 MHz-independent: s_memtime = shader clock)
  const double simd_cyc_per_iter = per_iter / waves_per_simd;     // SIMD cycles per wave-iteration
  printf("%-34s waves/SIMD %d  VALU %3d  MFMA %2d (%s)  | wave: %8.1f cycles/iter | SIMD: %7.1f cycles per wave-iteration = %5.2f cycles/VALU %s| %.3f ms\n",
         name, waves_per_simd, NV, NM, KIND ? "32x32x2" : "16x16x4", per_iter, simd_cyc_per_iter, NV ? simd_cyc_per_iter / NV : 0.0,
         NM ? "" : "", ms);
  if (NM) printf("%-34s                                      -> %6.1f SIMD cycles per MFMA\n", "", simd_cyc_per_iter / NM);
  hipFree(sink); hipFree(cyc);
}

int main() {
  for (int w : {1, 2, 4}) {
    run<64, 0, 0>("vector only", w);
    run<0, 8, 0>("matrix only 16x16x4", w);
    run<0, 8, 1>("matrix only 32x32x2", w);
    run<64, 2, 0>("vector + 2 x 16x16x4", w);
    run<64, 4, 0>("vector + 4 x 16x16x4", w);
    run<64, 8, 0>("vector + 8 x 16x16x4", w);
    run<64, 2, 1>("vector + 2 x 32x32x2", w);
    run<64, 4, 1>("vector + 4 x 32x32x2", w);
    printf("\n");
  }
  return 0;
}
