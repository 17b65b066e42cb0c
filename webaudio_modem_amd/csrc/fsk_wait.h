// fsk_wait.h -- the bound on every hand-off wait of the multi-wave kernels (demodulator: two, four, seven waves and the exact
// path's two; modulator: chain wave and owners).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Hand-off waits are bounded (include/fskhip.h, "Hand-off waits").  Every wait of the multi-wave kernels is a poll of an LDS
// counter with s_sleep in between; the producer is a wave of the same workgroup and always running, so a wait ends after at most
// a few tiles' work (a tile redone sample by sample: ~300 polls) -- unless a counter update were lost, which would be a hung GPU.
// A wave that has polled 2^FSK_SPIN_CAP_LOG2 times in ONE wait without getting on therefore sets bit 0 of the engine's hand-off
// fault word (DemodState::blk_stat[2]) and ends; the waves waiting on it run into their own bound, the launch finishes, and the
// host reports FSKHIP_E_HANDOFF (fskhip_synchronize / fskhip_get_faults / the _host calls / every later demodulate call).  2^22
// polls are >= 0.15 s of shader clock, four orders of magnitude beyond the longest wait of a healthy launch; polls are counted,
// not timed, so a preempted queue does not trip it.  All of it sits on the waits' slow paths: three scalar instructions and a
// short branch per FAILED poll, nothing on a step that finds its tile (measured: - 0.1 ... - 0.3 %, profiles/r06_handoff_bound.txt).
// `stat` must not be null (fskhip_create allocates the words for every engine).
#ifndef FSK_SPIN_CAP_LOG2
#define FSK_SPIN_CAP_LOG2 22
#endif
#define FSK_SPIN_CAP (1u << FSK_SPIN_CAP_LOG2)
#define FSK_WAIT_DECL uint32_t fsk_spins = 0;
#define FSK_WAIT_BEGIN fsk_spins = 0;
// One failed poll: sleep, count, and -- inside the SAME asm statement, so that the compiler's control flow graph is the unbounded
// loop's (a visible `if (...) { flag; s_endpgm }` at every wait cost 4.5 % at config #3 and 9 % in the seven-wave kernel: duplicated
// loops, twice the spill code; profiles/r06_handoff_bound.txt) -- at 2^FSK_SPIN_CAP_LOG2 polls: lane 0 ORs bit 0 into the fault
// word and the wave ends (it never returns to compiled code, so the two VGPRs it uses for that need no clobber).
#define FSK_SPIN_STR2(x) #x
#define FSK_SPIN_STR(x) FSK_SPIN_STR2(x)
#define FSK_SPIN(arg, stat)                                                                     \
  asm volatile("s_sleep %1\n\t"                                                                 \
               "s_add_u32 %0, %0, 1\n\t"                                                        \
               "s_bitcmp0_b32 %0, " FSK_SPIN_STR(FSK_SPIN_CAP_LOG2) "\n\t"                       \
               "s_cbranch_scc1 .Lfsk_spin_ok%=\n\t"                                             \
               "s_mov_b64 exec, 1\n\t"                                                          \
               "v_mov_b32 v0, 0\n\t"                                                            \
               "v_mov_b32 v1, 1\n\t"                                                            \
               "global_atomic_or v0, v1, %2 offset:8\n\t"                                       \
               "s_endpgm\n"                                                                     \
               ".Lfsk_spin_ok%=:"                                                                \
               : "+s"(fsk_spins)                                                                \
               : "n"(arg), "s"(stat)                                                            \
               : "scc", "memory")
