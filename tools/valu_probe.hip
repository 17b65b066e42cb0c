// valu_probe.hip -- what one vector instruction costs on gfx950 (MI355X), by kind, dependency and waves per SIMD.
//
// VERDICT r01 "weak #4": DESIGN.md argued "every VALU instruction occupies its SIMD for 4 cycles" without a
// measurement; the microarchitecture guide says 2 cycles for plain fp32 once >= 2 waves interleave, 8 for
// transcendentals, and calls packed fp32 an anti-lever.  This probe decides it on the part itself.
//
// Method: grid = one workgroup per CU (the LDS request keeps a second one out), 256*W threads = W waves on each
// of the CU's 4 SIMDs.  Every wave runs `iters` iterations of a 32-instruction inline-asm block (independent
// destinations, or one dependent chain) between two s_memtime stamps; a row reports
//   per_wave  = cycles one wave needs per instruction            (latency-bound when the chain is dependent)
//   per_simd  = per_wave / W = SIMD cycles per wave-instruction  (the throughput figure)
// plus the shader clock measured against s_memrealtime (100 MHz).
//
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/build/valu_probe tools/valu_probe.hip     Run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

struct Stamp { uint64_t cyc, rt; };

// 8 independent accumulators a0..a7 (and packed p0..p7); "dep" variants use a0/p0 only.
#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

#define IND8(op)                                                                                      \
  op("%0") op("%1") op("%2") op("%3") op("%4") op("%5") op("%6") op("%7")
#define IND32(op) IND8(op) IND8(op) IND8(op) IND8(op)

#define KERNEL_BEGIN(name)                                                                            \
  __global__ __launch_bounds__(1024) void name(Stamp *out, int iters, float seed) {                   \
    extern __shared__ float lds[];                                                                    \
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6,    \
          a7 = seed + 7;                                                                              \
    const float b = 0.999f, c = 0.001f;                                                               \
    f2 p0 = {seed, seed}, p1 = p0 + 1.f, p2 = p0 + 2.f, p3 = p0 + 3.f, p4 = p0 + 4.f, p5 = p0 + 5.f,  \
       p6 = p0 + 6.f, p7 = p0 + 7.f;                                                                  \
    const f2 pb = {0.999f, 0.998f}, pc = {0.001f, 0.002f};                                            \
    (void)lds; (void)b; (void)c; (void)pb; (void)pc;                                                  \
    __syncthreads();                                                                                  \
    const uint64_t t0 = __builtin_readcyclecounter();                                                 \
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();                                             \
    for (int it = 0; it < iters; it++) {

#define KERNEL_END                                                                                    \
    }                                                                                                 \
    const uint64_t t1 = __builtin_readcyclecounter();                                                 \
    const uint64_t r1 = __builtin_amdgcn_s_memrealtime();                                             \
    float sink = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + p4.x + p5.x + p6.x + p7.x + \
                 p0.y + p1.y + p2.y + p3.y + p4.y + p5.y + p6.y + p7.y;                               \
    if (sink == 123.456f) out[0].cyc = 0;                                                             \
    if ((threadIdx.x & 63) == 0) {                                                                    \
      const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                              \
      out[w].cyc = t1 - t0; out[w].rt = r1 - r0;                                                      \
    }                                                                                                 \
  }

#define S_REGS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)
#define P_REGS : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc)

// ---- plain fp32 -----------------------------------------------------------------------------------
#define OP_FMA(d) "v_fma_f32 " d ", " d ", %8, %9\n\t"
KERNEL_BEGIN(k_fma_ind) asm volatile(IND32(OP_FMA) S_REGS); KERNEL_END
KERNEL_BEGIN(k_fma_dep) asm volatile(REP32("v_fma_f32 %0, %0, %8, %9\n\t") S_REGS); KERNEL_END
// two interleaved dependent chains
KERNEL_BEGIN(k_fma_dep2) asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\t" "v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\t") S_REGS); KERNEL_END
#define OP_MUL(d) "v_mul_f32 " d ", " d ", %8\n\t"
KERNEL_BEGIN(k_mul_ind) asm volatile(IND32(OP_MUL) S_REGS); KERNEL_END
#define OP_FMAK(d) "v_fmaak_f32 " d ", " d ", %8, 0x3dc89600\n\t"
KERNEL_BEGIN(k_fmaak_ind) asm volatile(IND32(OP_FMAK) S_REGS); KERNEL_END
KERNEL_BEGIN(k_fmaak_dep) asm volatile(REP32("v_fmaak_f32 %0, %0, %8, 0x3dc89600\n\t") S_REGS); KERNEL_END
#define OP_MED3(d) "v_med3_f32 " d ", " d ", %8, %9\n\t"
KERNEL_BEGIN(k_med3_ind) asm volatile(IND32(OP_MED3) S_REGS); KERNEL_END
// ---- packed fp32 ----------------------------------------------------------------------------------
#define OP_PKFMA(d) "v_pk_fma_f32 " d ", " d ", %8, %9\n\t"
KERNEL_BEGIN(k_pkfma_ind) asm volatile(IND32(OP_PKFMA) P_REGS); KERNEL_END
// a dependent packed chain needs the wait state the assembler would insert (s_nop 0), as in the r01 kernel
KERNEL_BEGIN(k_pkfma_dep) asm volatile(REP32("v_pk_fma_f32 %0, %0, %8, %9\n\ts_nop 0\n\t") P_REGS); KERNEL_END
KERNEL_BEGIN(k_pkfma_dep_nonop) asm volatile(REP32("v_pk_fma_f32 %0, %0, %8, %9\n\t") P_REGS); KERNEL_END
#define OP_PKADD(d) "v_pk_add_f32 " d ", " d ", %8\n\t"
KERNEL_BEGIN(k_pkadd_ind) asm volatile(IND32(OP_PKADD) P_REGS); KERNEL_END
#define OP_PKMUL(d) "v_pk_mul_f32 " d ", " d ", %8\n\t"
KERNEL_BEGIN(k_pkmul_ind) asm volatile(IND32(OP_PKMUL) P_REGS); KERNEL_END
// ---- transcendental -------------------------------------------------------------------------------
#define OP_RCP(d) "v_rcp_f32 " d ", " d "\n\t"
KERNEL_BEGIN(k_rcp_ind) asm volatile(IND32(OP_RCP) S_REGS); KERNEL_END
KERNEL_BEGIN(k_rcp_dep) asm volatile(REP32("v_rcp_f32 %0, %0\n\t") S_REGS); KERNEL_END
#define OP_SQRT(d) "v_sqrt_f32 " d ", " d "\n\t"
KERNEL_BEGIN(k_sqrt_ind) asm volatile(IND32(OP_SQRT) S_REGS); KERNEL_END
#define OP_SIN(d) "v_sin_f32 " d ", " d "\n\t"
KERNEL_BEGIN(k_sin_ind) asm volatile(IND32(OP_SIN) S_REGS); KERNEL_END
#define OP_RSQ(d) "v_rsq_f32 " d ", " d "\n\t"
KERNEL_BEGIN(k_rsq_ind) asm volatile(IND32(OP_RSQ) S_REGS); KERNEL_END
// one transcendental per 8 instructions, rest independent fma (does the trans pipe overlap with the fma pipe?)
KERNEL_BEGIN(k_mix_rcp1_fma7) asm volatile(REP4("v_rcp_f32 %0, %0\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                                                 "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9\n\t") S_REGS); KERNEL_END
KERNEL_BEGIN(k_mix_rcp1_fma3) asm volatile(REP8("v_rcp_f32 %0, %0\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t") S_REGS); KERNEL_END
// rcp feeding a dependent fma (the AGC shape): rcp -> fma -> fma -> med3 -> mul -> (next)
KERNEL_BEGIN(k_agc_chain) asm volatile(REP8("v_mul_f32 %1, %0, %8\n\tv_rcp_f32 %2, %1\n\tv_fma_f32 %2, %2, %9, %0\n\tv_med3_f32 %0, %2, %8, %9\n\t") S_REGS); KERNEL_END
// ---- integer / select / compare -------------------------------------------------------------------
#define OP_ADDU(d) "v_add_u32 " d ", " d ", %8\n\t"
KERNEL_BEGIN(k_addu_ind) asm volatile(IND32(OP_ADDU) S_REGS); KERNEL_END
#define OP_BCNT(d) "v_bcnt_u32_b32 " d ", " d ", %8\n\t"
KERNEL_BEGIN(k_bcnt_ind) asm volatile(IND32(OP_BCNT) S_REGS); KERNEL_END
#define OP_CNDM(d) "v_cndmask_b32 " d ", " d ", %8, vcc\n\t"
KERNEL_BEGIN(k_cndmask_ind) asm volatile(IND32(OP_CNDM) S_REGS : "vcc"); KERNEL_END
KERNEL_BEGIN(k_cmp_cndmask) asm volatile(REP4("v_cmp_gt_f32 vcc, %0, %8\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_cmp_gt_f32 vcc, %2, %8\n\tv_cndmask_b32 %3, %3, %8, vcc\n\t"
                                              "v_cmp_gt_f32 vcc, %4, %8\n\tv_cndmask_b32 %5, %5, %8, vcc\n\tv_cmp_gt_f32 vcc, %6, %8\n\tv_cndmask_b32 %7, %7, %8, vcc\n\t") S_REGS : "vcc"); KERNEL_END
KERNEL_BEGIN(k_cmp_sgpr) asm volatile(REP8("v_cmp_gt_f32 s[20:21], %0, %8\n\tv_cmp_gt_f32 s[22:23], %1, %8\n\tv_cmp_gt_f32 s[24:25], %2, %8\n\tv_cmp_gt_f32 s[26:27], %3, %8\n\t") S_REGS : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27"); KERNEL_END
// ---- more shapes the demodulator uses ---------------------------------------------------------------
#define OP_MOVS(d) "v_mov_b32 " d ", s20\n\t"
KERNEL_BEGIN(k_mov_sgpr) asm volatile(IND32(OP_MOVS) S_REGS : "s20"); KERNEL_END
#define OP_FMA2(d) "v_fma_f32 " d ", 2.0, " d ", %8\n\t"
KERNEL_BEGIN(k_fma_inline2) asm volatile(IND32(OP_FMA2) S_REGS); KERNEL_END
#define OP_FMANEG(d) "v_fma_f32 " d ", -" d ", %8, |%9|\n\t"
KERNEL_BEGIN(k_fma_mods) asm volatile(IND32(OP_FMANEG) S_REGS); KERNEL_END
#define OP_MAX(d) "v_max_f32 " d ", " d ", %8\n\t"
KERNEL_BEGIN(k_max_ind) asm volatile(IND32(OP_MAX) S_REGS); KERNEL_END
#define OP_MAXABS(d) "v_max_f32_e64 " d ", |" d "|, |%8|\n\t"
KERNEL_BEGIN(k_maxabs_ind) asm volatile(IND32(OP_MAXABS) S_REGS); KERNEL_END
#define OP_MULLIT(d) "v_mul_f32 " d ", 0x3e22f983, " d "\n\t"
KERNEL_BEGIN(k_mul_literal) asm volatile(IND32(OP_MULLIT) S_REGS); KERNEL_END
#define OP_RNDNE(d) "v_rndne_f32 " d ", " d "\n\t"
KERNEL_BEGIN(k_rndne_ind) asm volatile(IND32(OP_RNDNE) S_REGS); KERNEL_END
#define OP_BFI(d) "v_bfi_b32 " d ", %8, " d ", %9\n\t"
KERNEL_BEGIN(k_bfi_ind) asm volatile(IND32(OP_BFI) S_REGS); KERNEL_END
#define OP_LSHLOR(d) "v_lshl_or_b32 " d ", " d ", 1, %8\n\t"
KERNEL_BEGIN(k_lshlor_ind) asm volatile(IND32(OP_LSHLOR) S_REGS); KERNEL_END
#define OP_AND(d) "v_and_b32 " d ", " d ", %8\n\t"
KERNEL_BEGIN(k_and_ind) asm volatile(IND32(OP_AND) S_REGS); KERNEL_END
#define OP_BITOP3(d) "v_bitop3_b32 " d ", " d ", %8, %9 bitop3:0x48\n\t"
KERNEL_BEGIN(k_bitop3_ind) asm volatile(IND32(OP_BITOP3) S_REGS); KERNEL_END
#define OP_ADDC(d) "v_addc_co_u32 " d ", vcc, " d ", %8, vcc\n\t"
KERNEL_BEGIN(k_addc_ind) asm volatile(IND32(OP_ADDC) S_REGS : "vcc"); KERNEL_END
#define OP_CMPVCC(d) "v_cmp_gt_f32 vcc, " d ", %8\n\t"
KERNEL_BEGIN(k_cmp_vcc) asm volatile(IND32(OP_CMPVCC) S_REGS : "vcc"); KERNEL_END
#define OP_CNDS(d) "v_cndmask_b32 " d ", " d ", %8, s[20:21]\n\t"
KERNEL_BEGIN(k_cndmask_sgprmask) asm volatile(IND32(OP_CNDS) S_REGS : "s20", "s21"); KERNEL_END
#define OP_CMPU(d) "v_cmp_eq_u32 vcc, s20, " d "\n\t"
KERNEL_BEGIN(k_cmp_u32_sgpr) asm volatile(IND32(OP_CMPU) S_REGS : "vcc", "s20"); KERNEL_END
#define OP_CVT(d) "v_cvt_f32_u32 " d ", " d "\n\t"
KERNEL_BEGIN(k_cvt_ind) asm volatile(IND32(OP_CVT) S_REGS); KERNEL_END
KERNEL_BEGIN(k_snop0_fma) asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n\ts_nop 0\n\tv_fma_f32 %1, %1, %8, %9\n\ts_nop 0\n\t") S_REGS); KERNEL_END
KERNEL_BEGIN(k_waitcnt_fma) asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n\ts_waitcnt lgkmcnt(0)\n\tv_fma_f32 %1, %1, %8, %9\n\ts_waitcnt vmcnt(0)\n\t") S_REGS); KERNEL_END
KERNEL_BEGIN(k_dswrite_b128)
  { typedef float v4 __attribute__((ext_vector_type(4))); const v4 v = {a0, a1, a2, a3}; const uint32_t addr = (threadIdx.x & 63) * 16u + (threadIdx.x >> 6) * 1024u;
    asm volatile(REP8("ds_write_b128 %1, %0\n\t") "s_waitcnt lgkmcnt(0)\n\t" : : "v"(v), "v"(addr) : "memory"); }
KERNEL_END
KERNEL_BEGIN(k_dswrite_b32)
  { const uint32_t addr = (threadIdx.x & 63) * 4u + (threadIdx.x >> 6) * 256u;
    asm volatile(REP8("ds_write_b32 %1, %0\n\t") "s_waitcnt lgkmcnt(0)\n\t" : : "v"(a0), "v"(addr) : "memory"); }
KERNEL_END
// LDS reads issued among VALU work with one wait per 8: what the consumer side of an LDS hand-off costs
KERNEL_BEGIN(k_dsread_fma_mix)
  { typedef float v4 __attribute__((ext_vector_type(4))); v4 v; const uint32_t addr = (threadIdx.x & 63) * 16u;
    asm volatile("ds_read_b128 %0, %9\n\t" REP8("v_fma_f32 %1, %1, %10, %11\n\tv_fma_f32 %2, %2, %10, %11\n\tv_fma_f32 %3, %3, %10, %11\n\t") "s_waitcnt lgkmcnt(0)\n\t"
                 : "=&v"(v), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(addr), "v"(b), "v"(c) : "memory"); a7 += v.x; }
KERNEL_END

// ---- second batch: which encodings are FMA-class? -----------------------------------------------
#define OP_K_LSHLREV(d) "v_lshlrev_b32 " d ", 1, " d "\n\t"
KERNEL_BEGIN(k_lshlrev) asm volatile(IND32(OP_K_LSHLREV) S_REGS); KERNEL_END
#define OP_K_LSHRREV(d) "v_lshrrev_b32 " d ", 31, " d "\n\t"
KERNEL_BEGIN(k_lshrrev) asm volatile(IND32(OP_K_LSHRREV) S_REGS); KERNEL_END
#define OP_K_ASHRREV(d) "v_ashrrev_i32 " d ", 31, " d "\n\t"
KERNEL_BEGIN(k_ashrrev) asm volatile(IND32(OP_K_ASHRREV) S_REGS); KERNEL_END
#define OP_K_XOR(d) "v_xor_b32 " d ", " d ", %8\n\t"
KERNEL_BEGIN(k_xor) asm volatile(IND32(OP_K_XOR) S_REGS); KERNEL_END
#define OP_K_OR(d) "v_or_b32 " d ", " d ", %8\n\t"
KERNEL_BEGIN(k_or) asm volatile(IND32(OP_K_OR) S_REGS); KERNEL_END
#define OP_K_SUBF(d) "v_sub_f32 " d ", " d ", %8\n\t"
KERNEL_BEGIN(k_subf) asm volatile(IND32(OP_K_SUBF) S_REGS); KERNEL_END
#define OP_K_ADDF(d) "v_add_f32 " d ", " d ", %8\n\t"
KERNEL_BEGIN(k_addf) asm volatile(IND32(OP_K_ADDF) S_REGS); KERNEL_END
#define OP_K_MOVV(d) "v_mov_b32 " d ", %8\n\t"
KERNEL_BEGIN(k_movv) asm volatile(IND32(OP_K_MOVV) S_REGS); KERNEL_END
#define OP_K_ANDOR(d) "v_and_or_b32 " d ", " d ", %8, %9\n\t"
KERNEL_BEGIN(k_andor) asm volatile(IND32(OP_K_ANDOR) S_REGS); KERNEL_END
#define OP_K_ADD3(d) "v_add3_u32 " d ", " d ", %8, %9\n\t"
KERNEL_BEGIN(k_add3) asm volatile(IND32(OP_K_ADD3) S_REGS); KERNEL_END
#define OP_K_LSHLADD(d) "v_lshl_add_u32 " d ", " d ", 1, %8\n\t"
KERNEL_BEGIN(k_lshladd) asm volatile(IND32(OP_K_LSHLADD) S_REGS); KERNEL_END
#define OP_K_ALIGNBIT(d) "v_alignbit_b32 " d ", " d ", %8, 31\n\t"
KERNEL_BEGIN(k_alignbit) asm volatile(IND32(OP_K_ALIGNBIT) S_REGS); KERNEL_END
#define OP_K_MINF(d) "v_min_f32 " d ", " d ", %8\n\t"
KERNEL_BEGIN(k_minf) asm volatile(IND32(OP_K_MINF) S_REGS); KERNEL_END
#define OP_K_SUBU(d) "v_sub_u32 " d ", " d ", %8\n\t"
KERNEL_BEGIN(k_subu) asm volatile(IND32(OP_K_SUBU) S_REGS); KERNEL_END
#define OP_K_SUBREV(d) "v_subrev_u32 " d ", " d ", %8\n\t"
KERNEL_BEGIN(k_subrev) asm volatile(IND32(OP_K_SUBREV) S_REGS); KERNEL_END
#define OP_K_MULLO(d) "v_mul_lo_u32 " d ", " d ", %8\n\t"
KERNEL_BEGIN(k_mullo) asm volatile(IND32(OP_K_MULLO) S_REGS); KERNEL_END
#define OP_K_MULU24(d) "v_mul_u32_u24 " d ", " d ", %8\n\t"
KERNEL_BEGIN(k_mulu24) asm volatile(IND32(OP_K_MULU24) S_REGS); KERNEL_END
#define OP_K_MADU24(d) "v_mad_u32_u24 " d ", " d ", %8, %9\n\t"
KERNEL_BEGIN(k_madu24) asm volatile(IND32(OP_K_MADU24) S_REGS); KERNEL_END
#define OP_K_BFE(d) "v_bfe_u32 " d ", " d ", 3, 5\n\t"
KERNEL_BEGIN(k_bfe) asm volatile(IND32(OP_K_BFE) S_REGS); KERNEL_END
#define OP_K_PERM(d) "v_perm_b32 " d ", " d ", %8, %9\n\t"
KERNEL_BEGIN(k_perm) asm volatile(IND32(OP_K_PERM) S_REGS); KERNEL_END
#define OP_K_MULLEGACY(d) "v_mul_legacy_f32 " d ", " d ", %8\n\t"
KERNEL_BEGIN(k_mullegacy) asm volatile(IND32(OP_K_MULLEGACY) S_REGS); KERNEL_END
#define OP_K_LDEXP(d) "v_ldexp_f32 " d ", " d ", 1\n\t"
KERNEL_BEGIN(k_ldexp) asm volatile(IND32(OP_K_LDEXP) S_REGS); KERNEL_END
#define OP_K_FLOOR(d) "v_floor_f32 " d ", " d "\n\t"
KERNEL_BEGIN(k_floor) asm volatile(IND32(OP_K_FLOOR) S_REGS); KERNEL_END
#define OP_K_FRACT(d) "v_fract_f32 " d ", " d "\n\t"
KERNEL_BEGIN(k_fract) asm volatile(IND32(OP_K_FRACT) S_REGS); KERNEL_END
#define OP_K_FMAC(d) "v_fmac_f32 " d ", %8, %9\n\t"
KERNEL_BEGIN(k_fmac) asm volatile(IND32(OP_K_FMAC) S_REGS); KERNEL_END
#define OP_K_MULCLAMP(d) "v_mul_f32_e64 " d ", " d ", %8 clamp\n\t"
KERNEL_BEGIN(k_mulclamp) asm volatile(IND32(OP_K_MULCLAMP) S_REGS); KERNEL_END
#define OP_K_FMACLAMP(d) "v_fma_f32 " d ", " d ", %8, %9 clamp\n\t"
KERNEL_BEGIN(k_fmaclamp) asm volatile(IND32(OP_K_FMACLAMP) S_REGS); KERNEL_END
#define OP_K_MAXU(d) "v_max_u32 " d ", " d ", %8\n\t"
KERNEL_BEGIN(k_maxu) asm volatile(IND32(OP_K_MAXU) S_REGS); KERNEL_END
#define OP_K_CMPCLASS(d) "v_cmp_class_f32 vcc, " d ", %8\n\t"
KERNEL_BEGIN(k_cmpclass) asm volatile(IND32(OP_K_CMPCLASS) S_REGS : "vcc"); KERNEL_END
#define OP_K_SAD(d) "v_sad_u8 " d ", " d ", %8, %9\n\t"
KERNEL_BEGIN(k_sad) asm volatile(IND32(OP_K_SAD) S_REGS); KERNEL_END
#define OP_K_DOT2(d) "v_dot2c_f32_f16 " d ", %8, %9\n\t"
KERNEL_BEGIN(k_dot2) asm volatile(IND32(OP_K_DOT2) S_REGS); KERNEL_END

// ---- scalar / nop ---------------------------------------------------------------------------------
KERNEL_BEGIN(k_snop0) asm volatile(REP32("s_nop 0\n\t") S_REGS); KERNEL_END
KERNEL_BEGIN(k_salu) asm volatile(REP32("s_add_u32 s20, s20, 1\n\t") S_REGS : "s20", "scc"); KERNEL_END
// VALU and SALU alternating: does a wave dual-issue them? (cost vs k_fma_ind)
KERNEL_BEGIN(k_fma_salu) asm volatile(REP4("v_fma_f32 %0, %0, %8, %9\n\ts_add_u32 s20, s20, 1\n\tv_fma_f32 %1, %1, %8, %9\n\ts_add_u32 s21, s21, 1\n\t"
                                           "v_fma_f32 %2, %2, %8, %9\n\ts_add_u32 s20, s20, 1\n\tv_fma_f32 %3, %3, %8, %9\n\ts_add_u32 s21, s21, 1\n\t") S_REGS : "s20", "s21", "scc"); KERNEL_END
// VOP3 with an SGPR source (constants in SGPRs, as the UNI kernels use)
KERNEL_BEGIN(k_fma_sgpr) asm volatile(IND32(OP_FMA) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(b), "v"(c)); KERNEL_END
// ---- LDS ------------------------------------------------------------------------------------------
KERNEL_BEGIN(k_dsread_b128)
  { typedef float v4 __attribute__((ext_vector_type(4))); v4 v; const uint32_t addr = (threadIdx.x & 63) * 16u;
    asm volatile(REP8("ds_read_b128 %0, %1\n\t") "s_waitcnt lgkmcnt(0)\n\t" : "=&v"(v) : "v"(addr) : "memory"); a0 += v.x; }
KERNEL_END
KERNEL_BEGIN(k_dsread_b32)
  { float v; const uint32_t addr = (threadIdx.x & 63) * 4u;
    asm volatile(REP8("ds_read_b32 %0, %1\n\t") "s_waitcnt lgkmcnt(0)\n\t" : "=&v"(v) : "v"(addr) : "memory"); a0 += v; }
KERNEL_END


// ---- third batch (round 3): what the back wave's per-sample control flow and memory operations cost a wave ----------
// 8 independent v_fma as the background, plus ONE instance of the shape under test per 8: the difference to 9 plain fma
// is the shape's price.
#define FMA8 "v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t" \
             "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9\n\t"
#define FMA8X "v_fma_f32 %0, %0, %9, %10\n\tv_fma_f32 %1, %1, %9, %10\n\tv_fma_f32 %2, %2, %9, %10\n\tv_fma_f32 %3, %3, %9, %10\n\t" \
              "v_fma_f32 %4, %4, %9, %10\n\tv_fma_f32 %5, %5, %9, %10\n\tv_fma_f32 %6, %6, %9, %10\n\tv_fma_f32 %7, %7, %9, %10\n\t"
#define RD_REGS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(v) : "v"(b), "v"(c), "v"(addr) : "memory"
KERNEL_BEGIN(k_bg_fma8) asm volatile(REP4(FMA8) S_REGS); KERNEL_END
// v_cmp -> vcc -> s_cbranch_vccnz, never taken (the rare-event test of the frame logic)
KERNEL_BEGIN(k_br_vcc) asm volatile(REP4(FMA8 "v_cmp_gt_f32 vcc, 0, %8\n\ts_cbranch_vccnz 1f\n\t") "1:\n\t" S_REGS : "vcc"); KERNEL_END
// v_cmp -> SGPR pair -> s_and_b64 vcc -> s_cbranch_vccz, always taken to the next line (the start/stop-bit test)
KERNEL_BEGIN(k_br_sand) asm volatile(REP4(FMA8 "v_cmp_gt_f32 s[20:21], 0, %8\n\tv_cmp_gt_f32 vcc, 0, %9\n\ts_and_b64 vcc, s[20:21], vcc\n\ts_cbranch_vccnz 1f\n\t") "1:\n\t" S_REGS : "vcc", "s20", "s21", "scc"); KERNEL_END
// a taken branch (jump over one instruction)
KERNEL_BEGIN(k_br_taken) asm volatile(REP4(FMA8 "s_cbranch_vccz 2f\n\ts_nop 0\n\t2:\n\t") S_REGS : "vcc"); KERNEL_END
// scalar compare + branch, never taken
KERNEL_BEGIN(k_br_scc) asm volatile(REP4(FMA8 "s_cmp_eq_u32 s20, 12345\n\ts_cbranch_scc1 1f\n\t") "1:\n\t" S_REGS : "scc", "s20"); KERNEL_END
// v_readfirstlane -> s_cmp -> branch (a wave-uniform decision taken from a VGPR)
KERNEL_BEGIN(k_br_rfl) asm volatile(REP4(FMA8 "v_readfirstlane_b32 s20, %0\n\ts_cmp_eq_u32 s20, 12345\n\ts_cbranch_scc1 1f\n\t") "1:\n\t" S_REGS : "scc", "s20"); KERNEL_END
// one global store of a dword per lane (the amplitude ring), never waited for
KERNEL_BEGIN(k_gstore)
  { const uint64_t addr = reinterpret_cast<uint64_t>(out) + (1u << 20) + ((uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 4u);
    asm volatile(REP4(FMA8 "global_store_dword %10, %0, off\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "v"(addr) : "memory"); }
KERNEL_END
// LDS: write a dword / read a dword and wait for it at once (exposed latency) / read and wait 8 instructions later
KERNEL_BEGIN(k_ldsw32)
  { const uint32_t addr = threadIdx.x * 4u;
    asm volatile(REP4(FMA8 "ds_write_b32 %10, %0\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "v"(addr) : "memory"); }
KERNEL_END
KERNEL_BEGIN(k_ldsr32_wait)
  { const uint32_t addr = threadIdx.x * 4u; float v;
    asm volatile(REP4(FMA8X "ds_read_b32 %8, %11\n\ts_waitcnt lgkmcnt(0)\n\t") RD_REGS); a0 += v; }
KERNEL_END
KERNEL_BEGIN(k_ldsr128_wait)
  { typedef float v4 __attribute__((ext_vector_type(4))); const uint32_t addr = threadIdx.x * 16u; v4 v;
    asm volatile(REP4(FMA8X "ds_read_b128 %8, %11\n\ts_waitcnt lgkmcnt(0)\n\t") RD_REGS); a0 += v.x; }
KERNEL_END
KERNEL_BEGIN(k_ldsr128_late)
  { typedef float v4 __attribute__((ext_vector_type(4))); const uint32_t addr = threadIdx.x * 16u; v4 v;
    asm volatile(REP4("ds_read_b128 %8, %11\n\t" FMA8X "s_waitcnt lgkmcnt(0)\n\t") RD_REGS); a0 += v.x; }
KERNEL_END
// the hand-off poll as the kernels do it: ds_read_b32 -> wait -> v_readfirstlane
KERNEL_BEGIN(k_lds_peek)
  { const uint32_t addr = 0; float v;
    asm volatile(REP4(FMA8X "ds_read_b32 %8, %11\n\ts_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 s20, %8\n\t") RD_REGS, "s20"); a0 += v; }
KERNEL_END
// s_sleep 1
KERNEL_BEGIN(k_sleep1) asm volatile(REP4(FMA8 "s_sleep 1\n\t") S_REGS); KERNEL_END

typedef void (*kern_t)(Stamp *, int, float);
struct Row { const char *name; kern_t k; int per_iter; const char *note; };

int main(int argc, char **argv) {
  const char *only = argc > 1 ? argv[1] : nullptr;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("# device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
  const Row rows[] = {
      {"v_fma_f32 independent", k_fma_ind, 32, ""},
      {"v_fma_f32 dependent chain", k_fma_dep, 32, ""},
      {"v_fma_f32 two dependent chains", k_fma_dep2, 32, ""},
      {"v_mul_f32 (VOP2) independent", k_mul_ind, 32, ""},
      {"v_fmaak_f32 (64-bit literal) independent", k_fmaak_ind, 32, ""},
      {"v_fmaak_f32 dependent (polynomial shape)", k_fmaak_dep, 32, ""},
      {"v_fma_f32 with SGPR source", k_fma_sgpr, 32, ""},
      {"v_med3_f32 independent", k_med3_ind, 32, ""},
      {"v_pk_fma_f32 independent", k_pkfma_ind, 32, "2 lanes-ops each"},
      {"v_pk_fma_f32 dependent + s_nop 0", k_pkfma_dep, 32, "per pk instruction (nop included)"},
      {"v_pk_fma_f32 dependent, no nop", k_pkfma_dep_nonop, 32, "hazard unpadded: timing only"},
      {"v_pk_add_f32 independent", k_pkadd_ind, 32, ""},
      {"v_pk_mul_f32 independent", k_pkmul_ind, 32, ""},
      {"v_rcp_f32 independent", k_rcp_ind, 32, ""},
      {"v_rcp_f32 dependent", k_rcp_dep, 32, ""},
      {"v_sqrt_f32 independent", k_sqrt_ind, 32, ""},
      {"v_rsq_f32 independent", k_rsq_ind, 32, ""},
      {"v_sin_f32 independent", k_sin_ind, 32, ""},
      {"1 v_rcp + 7 v_fma independent", k_mix_rcp1_fma7, 32, "per instruction of the mix"},
      {"1 v_rcp + 3 v_fma independent", k_mix_rcp1_fma3, 32, "per instruction of the mix"},
      {"AGC-shaped chain mul>rcp>fma>med3", k_agc_chain, 32, "per instruction; x4 = per AGC step"},
      {"v_add_u32 independent", k_addu_ind, 32, ""},
      {"v_bcnt_u32_b32 independent", k_bcnt_ind, 32, ""},
      {"v_cndmask_b32 (vcc) independent", k_cndmask_ind, 32, ""},
      {"v_cmp>vcc + v_cndmask pairs", k_cmp_cndmask, 32, "per instruction"},
      {"v_cmp_gt_f32 -> SGPR pair", k_cmp_sgpr, 32, ""},
      {"v_mov_b32 from SGPR", k_mov_sgpr, 32, ""},
      {"v_fma_f32 with inline constant 2.0", k_fma_inline2, 32, ""},
      {"v_fma_f32 with neg/abs modifiers", k_fma_mods, 32, ""},
      {"v_max_f32 (VOP2)", k_max_ind, 32, ""},
      {"v_max_f32 |a|,|b| (VOP3 modifiers)", k_maxabs_ind, 32, ""},
      {"v_mul_f32 with 32-bit literal", k_mul_literal, 32, ""},
      {"v_rndne_f32", k_rndne_ind, 32, ""},
      {"v_bfi_b32", k_bfi_ind, 32, ""},
      {"v_lshl_or_b32", k_lshlor_ind, 32, ""},
      {"v_and_b32 (VOP2)", k_and_ind, 32, ""},
      {"v_bitop3_b32", k_bitop3_ind, 32, ""},
      {"v_addc_co_u32 (vcc in/out)", k_addc_ind, 32, ""},
      {"v_cmp_gt_f32 -> vcc", k_cmp_vcc, 32, ""},
      {"v_cndmask_b32 with SGPR-pair mask", k_cndmask_sgprmask, 32, ""},
      {"v_cmp_eq_u32 vcc, SGPR, v", k_cmp_u32_sgpr, 32, ""},
      {"v_cvt_f32_u32", k_cvt_ind, 32, ""},
      {"v_fma + s_nop 0 alternating", k_snop0_fma, 32, "per instruction (16 VALU + 16 s_nop)"},
      {"v_fma + s_waitcnt alternating", k_waitcnt_fma, 32, "per instruction (16 VALU + 16 s_waitcnt, nothing pending)"},
      {"ds_write_b128 x8 + wait", k_dswrite_b128, 8, "per ds_write (1 KiB/wave)"},
      {"ds_write_b32 x8 + wait", k_dswrite_b32, 8, "per ds_write"},
      {"ds_read_b128 + 24 v_fma + wait", k_dsread_fma_mix, 25, "per instruction"},
      {"v_lshlrev_b32 v, 1, v", k_lshlrev, 32, ""},
      {"v_lshrrev_b32 v, 31, v", k_lshrrev, 32, ""},
      {"v_ashrrev_i32 v, 31, v", k_ashrrev, 32, ""},
      {"v_xor_b32", k_xor, 32, ""},
      {"v_or_b32", k_or, 32, ""},
      {"v_sub_f32", k_subf, 32, ""},
      {"v_add_f32", k_addf, 32, ""},
      {"v_mov_b32 VGPR", k_movv, 32, ""},
      {"v_and_or_b32", k_andor, 32, ""},
      {"v_add3_u32", k_add3, 32, ""},
      {"v_lshl_add_u32", k_lshladd, 32, ""},
      {"v_alignbit_b32", k_alignbit, 32, ""},
      {"v_min_f32", k_minf, 32, ""},
      {"v_sub_u32", k_subu, 32, ""},
      {"v_subrev_u32", k_subrev, 32, ""},
      {"v_mul_lo_u32", k_mullo, 32, ""},
      {"v_mul_u32_u24", k_mulu24, 32, ""},
      {"v_mad_u32_u24", k_madu24, 32, ""},
      {"v_bfe_u32", k_bfe, 32, ""},
      {"v_perm_b32", k_perm, 32, ""},
      {"v_mul_legacy_f32", k_mullegacy, 32, ""},
      {"v_ldexp_f32", k_ldexp, 32, ""},
      {"v_floor_f32", k_floor, 32, ""},
      {"v_fract_f32", k_fract, 32, ""},
      {"v_fmac_f32 (VOP2)", k_fmac, 32, ""},
      {"v_mul_f32 clamp", k_mulclamp, 32, ""},
      {"v_fma_f32 clamp", k_fmaclamp, 32, ""},
      {"v_max_u32", k_maxu, 32, ""},
      {"v_cmp_class_f32", k_cmpclass, 32, ""},
      {"v_sad_u8", k_sad, 32, ""},
      {"v_dot2c_f32_f16", k_dot2, 32, ""},
      {"s_nop 0", k_snop0, 32, ""},
      {"s_add_u32 dependent", k_salu, 32, ""},
      {"v_fma + s_add alternating", k_fma_salu, 32, "per instruction (16 VALU + 16 SALU)"},
      {"ds_read_b128 x8 + wait", k_dsread_b128, 8, "per ds_read (conflict-free, 1 KiB/wave)"},
      {"ds_read_b32 x8 + wait", k_dsread_b32, 8, "per ds_read"},
      {"r3 background: 8 v_fma", k_bg_fma8, 32, "per instruction"},
      {"r3 8 fma + v_cmp>vcc + s_cbranch_vccnz (not taken)", k_br_vcc, 4, "per GROUP of 8 fma + shape"},
      {"r3 8 fma + 2 v_cmp + s_and_b64 + s_cbranch (not taken)", k_br_sand, 4, "per group"},
      {"r3 8 fma + taken branch over one s_nop", k_br_taken, 4, "per group"},
      {"r3 8 fma + s_cmp + s_cbranch_scc1 (not taken)", k_br_scc, 4, "per group"},
      {"r3 8 fma + v_readfirstlane + s_cmp + s_cbranch", k_br_rfl, 4, "per group"},
      {"r3 8 fma + global_store_dword", k_gstore, 4, "per group"},
      {"r3 8 fma + ds_write_b32", k_ldsw32, 4, "per group"},
      {"r3 8 fma + ds_read_b32 + wait", k_ldsr32_wait, 4, "per group"},
      {"r3 8 fma + ds_read_b128 + wait", k_ldsr128_wait, 4, "per group"},
      {"r3 ds_read_b128, 8 fma, then wait", k_ldsr128_late, 4, "per group"},
      {"r3 8 fma + lds peek (read, wait, readfirstlane)", k_lds_peek, 4, "per group"},
      {"r3 8 fma + s_sleep 1", k_sleep1, 4, "per group"},
  };
  const int iters = 4096;
  Stamp *d_out;
  CHECK(hipMalloc(&d_out, (2u << 20) + (size_t)cus * 1024 * 4));   // stamps + the store target of k_gstore
  std::vector<Stamp> h(cus * 32);
  printf("%-44s %5s %10s %10s %8s  %s\n", "instruction pattern", "W", "per_wave", "per_simd", "GHz", "note");
  for (const Row &r : rows) {
    if (only && !strstr(r.name, only)) continue;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(r.k), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    for (int W : {1, 2, 3, 4}) {
      const int threads = 256 * W;
      const size_t lds = 96 * 1024;  // > half of 160 KiB: one workgroup per CU
      for (int rep = 0; rep < 2; rep++) {  // first pass warms the clock
        hipLaunchKernelGGL(r.k, dim3(cus), dim3(threads), lds, 0, d_out, iters, 1.0f);
        CHECK(hipDeviceSynchronize());
      }
      const int nw = cus * 4 * W;
      CHECK(hipMemcpy(h.data(), d_out, sizeof(Stamp) * nw, hipMemcpyDeviceToHost));
      std::vector<double> cyc(nw), ghz(nw);
      for (int i = 0; i < nw; i++) { cyc[i] = (double)h[i].cyc; ghz[i] = h[i].rt ? (double)h[i].cyc / ((double)h[i].rt * 10.0) : 0.0; }
      std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
      const double n_inst = (double)iters * r.per_iter;
      const double per_wave = cyc[nw / 2] / n_inst;
      printf("%-44s %5d %10.3f %10.3f %8.3f  %s\n", r.name, W, per_wave, per_wave / W, ghz[nw / 2], r.note);
      fflush(stdout);
    }
  }
  CHECK(hipFree(d_out));
  return 0;
}
