// read_probe.hip -- how fast can one wave per 64 rows stream a stream-major [rows][samples] float buffer?
// (measurement tooling, round 3: the two-wave demodulator with all its arithmetic removed still needs 74 % of its time,
// so the question is what the access pattern and the bytes in flight allow by themselves)
//
//   reg<CH16, DEPTH>   register staging: a visit = CH16 buffer_load_dwordx4 (64 rows x CH16*16 bytes), DEPTH visits in flight
//   dma<CH16, DEPTH>   the same visits landed by LDS-DMA (buffer_load_dwordx4 ... lds) in a ring of DEPTH slots, optionally
//                      read back by the lane that owns the row (ds_read_b128)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/build/read_probe tools/read_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

template <int N> __device__ inline void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

// ---- register staging ------------------------------------------------------------------------------------------
template <int CH16, int DEPTH>
__global__ __launch_bounds__(64) void reg_probe(const float *__restrict__ buf, size_t n, size_t pitch, float *__restrict__ sink) {
  constexpr int R = 64 / CH16;                      // rows per instruction
  const uint32_t lane = threadIdx.x;
  v4i rsrc;
  {
    const uint64_t base = reinterpret_cast<uint64_t>(buf + (size_t)blockIdx.x * 64u * pitch);
    rsrc.x = (int)(uint32_t)base; rsrc.y = (int)(uint32_t)(base >> 32); rsrc.z = (int)(uint32_t)(64u * pitch * 4u); rsrc.w = 0x00020000;
  }
  const uint32_t voff = (uint32_t)(((lane / CH16) * pitch + 4u * (lane % CH16)) * 4u);
  const uint32_t rowstep = (uint32_t)(R * pitch * 4u);
  const uint32_t nvis = (uint32_t)(n / (4u * CH16));
  v4f r[DEPTH][CH16];
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  auto issue = [&](uint32_t v, v4f (&dst)[CH16]) {
    const uint32_t vv = v < nvis ? v : nvis - 1;
#pragma unroll
    for (int i = 0; i < CH16; i++)
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst[i]) : "v"(voff + (uint32_t)i * rowstep), "s"(rsrc), "s"(vv * CH16 * 16u) : "memory");
  };
#pragma unroll
  for (int d = 0; d < DEPTH; d++) issue(d, r[d]);
  for (uint32_t v = 0; v < nvis; v += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
      wait_vmcnt<(DEPTH - 1) * CH16>();
#pragma unroll
      for (int i = 0; i < CH16; i++) { asm volatile("" : "+v"(r[d][i])); acc += r[d][i]; }
      issue(v + d + DEPTH, r[d]);
    }
  }
  wait_vmcnt<0>();
#pragma unroll
  for (int d = 0; d < DEPTH; d++)
#pragma unroll
    for (int i = 0; i < CH16; i++) asm volatile("" : "+v"(r[d][i]));
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

// ---- LDS-DMA -----------------------------------------------------------------------------------------------------
// Instruction i covers rows [i*R, i*R + R); lane l = r*CH16 + j fetches row i*R + r, piece j ^ s(r) (four adjacent lanes still
// cover one 64-byte line) and lands at slot + i*1024 + l*16.  Lane L (row L) later reads its piece c at
// slot + (L / R)*1024 + ((L % R)*CH16 + (c ^ s(L % R)))*16; s(r) = r / 4 makes that ds_read_b128 conflict-free for CH16 = 4.
template <int CH16, int DEPTH, int CONSUME, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void dma_probe(const float *__restrict__ buf, size_t n, size_t pitch, float *__restrict__ sink) {
  extern __shared__ float4 lds[];
  constexpr int R = 64 / CH16;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t group = blockIdx.x * WAVES + wave;
  v4i rsrc;
  {
    const uint64_t base = reinterpret_cast<uint64_t>(buf + (size_t)group * 64u * pitch);
    rsrc.x = (int)(uint32_t)base; rsrc.y = (int)(uint32_t)(base >> 32); rsrc.z = (int)(uint32_t)(64u * pitch * 4u); rsrc.w = 0x00020000;
    rsrc.x = __builtin_amdgcn_readfirstlane(rsrc.x); rsrc.y = __builtin_amdgcn_readfirstlane(rsrc.y);
  }
  const uint32_t swz = CH16 == 4 ? ((lane / CH16) / 4u) : 0u;
  const uint32_t voff = (uint32_t)(((lane / CH16) * pitch + 4u * ((lane % CH16) ^ swz)) * 4u);
  const uint32_t rowstep = (uint32_t)(R * pitch * 4u);
  const uint32_t nvis = (uint32_t)(n / (4u * CH16));
  const uint32_t rswz = CH16 == 4 ? ((lane % R) / 4u) : 0u;
  const uint32_t lds0 = (uint32_t)(uintptr_t)lds + wave * (uint32_t)(DEPTH * CH16 * 1024);
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  auto issue = [&](uint32_t v, uint32_t slot) {
    const uint32_t vv = v < nvis ? v : nvis - 1;
    const uint32_t m0 = __builtin_amdgcn_readfirstlane(lds0 + slot * (uint32_t)(CH16 * 1024));
    const uint32_t so = vv * CH16 * 16u;
#pragma unroll
    for (int i = 0; i < CH16; i++)
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                   : : "s"(m0 + (uint32_t)i * 1024u), "v"(voff + (uint32_t)i * rowstep), "s"(rsrc), "s"(so) : "memory");
  };
#pragma unroll
  for (int d = 0; d < DEPTH; d++) issue(d, d);
  for (uint32_t v = 0; v < nvis; v += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
      wait_vmcnt<(DEPTH - 1) * CH16 < 63 ? (DEPTH - 1) * CH16 : 63>();
      if (CONSUME) {
        const float4 *slot = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(lds) + wave * (DEPTH * CH16 * 1024) + d * (CH16 * 1024));
#pragma unroll
        for (int c = 0; c < CH16; c++) {
          const float4 x = slot[(lane / R) * 64 + (lane % R) * CH16 + ((uint32_t)c ^ rswz)];
          acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      issue(v + d + DEPTH, d);
    }
  }
  wait_vmcnt<0>();
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

// ---- plain linear copy-read for reference --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void linear_probe(const float4 *__restrict__ buf, size_t n4, float *__restrict__ sink) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256u) {
    const float4 v = buf[i];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CHECK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main(int argc, char **argv) {
  const uint32_t S = argc > 1 ? (uint32_t)atoi(argv[1]) : 65536u;
  const size_t N = argc > 2 ? (size_t)atol(argv[2]) : 49152;      // floats per row (multiple of 512)
  const char *only = argc > 3 ? argv[3] : "";
  float *buf, *sink;
  const size_t bytes = (size_t)S * N * 4u;
  CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(buf, 0, bytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  printf("# %u rows x %zu floats = %.2f GB; one wave per 64 rows unless noted\n", S, N, bytes / 1e9);
  auto report = [&](const char *name, int in_flight_kb, float ms) {
    printf("%-34s %3d KB in flight/wave  %8.3f ms  %7.1f GB/s\n", name, in_flight_kb, ms, bytes / ms / 1e6);
    fflush(stdout);
  };
  auto run = [&](const char *name, int kb, auto &&launch) {
    if (only[0] && !strstr(name, only)) return;
    launch(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); launch(); launch(); CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1)); report(name, kb, time_ms(e0, e1) / 2);
  };
  run("linear float4 grid-stride", 0, [&] { hipLaunchKernelGGL(linear_probe, dim3(256 * 16), dim3(256), 0, 0, (const float4 *)buf, bytes / 16, sink); });
#define REG(C, D) run("reg  chunk " #C "x16B depth " #D, C * D, [&] { hipLaunchKernelGGL((reg_probe<C, D>), dim3(S / 64), dim3(64), 0, 0, buf, N, N, sink); });
  REG(4, 3) REG(4, 6) REG(4, 12) REG(8, 3) REG(8, 6) REG(16, 2) REG(16, 3)
#define DMA(C, D, CONS, W)                                                                                          \
  {                                                                                                                 \
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&dma_probe<C, D, CONS, W>), hipFuncAttributeMaxDynamicSharedMemorySize, C * D * 1024 * W)); \
    run("dma  chunk " #C "x16B depth " #D " consume " #CONS " waves/wg " #W, C * D,                                 \
        [&] { hipLaunchKernelGGL((dma_probe<C, D, CONS, W>), dim3(S / 64 / W), dim3(64 * W), C * D * 1024 * W, 0, buf, N, N, sink); }); \
  }
  DMA(4, 3, 0, 1) DMA(4, 3, 1, 1) DMA(4, 6, 0, 1) DMA(4, 8, 0, 1) DMA(4, 8, 1, 1)
  DMA(8, 3, 0, 1) DMA(8, 4, 0, 1) DMA(8, 4, 1, 1)
  DMA(16, 2, 0, 1) DMA(16, 2, 1, 1) DMA(32, 1, 0, 1)
  DMA(4, 8, 1, 2) DMA(8, 4, 1, 2) DMA(16, 2, 1, 2)
  return 0;
}
