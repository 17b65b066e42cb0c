// occ_probe.hip -- how many workgroups of a given shape does a gfx950 CU hold at once?  (round 6: why only three 320-thread
// workgroups of demod_blk5_kernel -- 96 VGPRs, 40 352 B of LDS -- were co-resident per CU where four 256-thread ones are)
// Every workgroup stamps s_memrealtime at its start, then spins ~300 us.  Workgroups that start within the first 100 us are the
// ones the device held at once.   build: hipcc --offload-arch=gfx950 -O2 -o occ_probe occ_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

template <int THREADS, int WPE, int VG>
__global__ __launch_bounds__(THREADS, WPE) void probe(unsigned long long *t0, int spin_ticks) {
  extern __shared__ float lds[];
  const unsigned long long s = __builtin_amdgcn_s_memrealtime();
  float acc[VG];
#pragma unroll
  for (int i = 0; i < VG; i++) acc[i] = (float)(threadIdx.x + i);
  if (threadIdx.x == 0) t0[blockIdx.x] = s;
  lds[threadIdx.x] = acc[0];
  while (__builtin_amdgcn_s_memrealtime() - s < (unsigned long long)spin_ticks) {
#pragma unroll
    for (int i = 0; i < VG; i++) acc[i] = acc[i] * 1.0001f + lds[(threadIdx.x + i) & 63];
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < VG; i++) r += acc[i];
  if (r == 12345.678f) t0[0] = 0;
}

template <int THREADS, int WPE, int VG>
void run(const char *name, size_t lds, int wgs) {
  unsigned long long *d;
  hipMalloc(&d, sizeof(unsigned long long) * wgs);
  hipFuncSetAttribute(reinterpret_cast<const void *>(&probe<THREADS, WPE, VG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncAttributes fa;
  hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&probe<THREADS, WPE, VG>));
  int api = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, reinterpret_cast<const void *>(&probe<THREADS, WPE, VG>), THREADS, lds);
  hipLaunchKernelGGL((probe<THREADS, WPE, VG>), dim3(wgs), dim3(THREADS), lds, 0, d, 30000);   // 100 MHz ticks: 300 us
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(wgs);
  hipMemcpy(h.data(), d, sizeof(unsigned long long) * wgs, hipMemcpyDeviceToHost);
  const unsigned long long first = *std::min_element(h.begin(), h.end());
  int early = 0;
  for (auto v : h) early += (v - first) < 10000ull;   // within 100 us
  printf("%-44s threads %3d  numRegs %3d  lds %6zu  api %d/CU  resident at once %4d of %d = %.2f per CU\n", name, THREADS, fa.numRegs, lds, api, early, wgs,
         early / 256.0);
  hipFree(d);
}

int main() {
  run<256, 4, 100>("256 thr, compiled for 4 waves/SIMD", 36192, 2048);
  run<256, 4, 100>("256 thr, 4 waves/SIMD, lds 40352", 40352, 2048);
  run<320, 5, 80>("320 thr, 5 waves/SIMD (<= 96 VGPRs)", 40352, 2048);
  run<320, 5, 80>("320 thr, 5 waves/SIMD, lds 36192", 36192, 2048);
  run<320, 5, 80>("320 thr, 5 waves/SIMD, lds 32768", 32768, 2048);
  run<320, 5, 80>("320 thr, 5 waves/SIMD, lds 1024", 1024, 2048);
  run<320, 6, 64>("320 thr, 6 waves/SIMD (<= 80 VGPRs), lds 40352", 40352, 2048);
  run<320, 8, 48>("320 thr, 8 waves/SIMD (<= 64 VGPRs), lds 40352", 40352, 2048);
  run<384, 6, 64>("384 thr, 6 waves/SIMD, lds 40352", 40352, 2048);
  run<512, 8, 48>("512 thr, 8 waves/SIMD, lds 40352", 40352, 2048);
  run<256, 5, 80>("256 thr, 5 waves/SIMD, lds 32000", 32000, 2048);
  run<256, 5, 80>("256 thr, 5 waves/SIMD, lds 1024", 1024, 2048);
  run<64, 5, 80>("64 thr, 5 waves/SIMD, lds 1024", 1024, 8192);
  return 0;
}
