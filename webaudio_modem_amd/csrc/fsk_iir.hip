// fsk_iir.hip -- batched generic IIRFilter (src/dsp/filters.ts:8-106) + FilterFactory.createIIR* (325-344), the IIR half of
// dsp/filters.ts as a component of its own (VERDICT r04 "missing" #1; inside the demodulator the same filter lives as
// three hard-wired biquads).
//
// Direct Form I of any order <= 8, exactly as IIRFilter.process() evaluates it: output = 0; output += b[i] * x[n-i] for
// i = 0 .. b.length-1, left to right; output -= a[i] * y[n-i] for i = 1 .. a.length-1 -- every product and every sum rounded
// on its own (this file is built -ffp-contract=off), in doubles on the parity path, so that processBuffer()'s Float32Array
// and process()'s doubles are the reference's bit for bit.  A recurrence per stream: one lane per stream, the coefficients
// (one set for the whole batch) ride in SGPRs, the histories x[n-1..n-8] / y[n-1..n-8] in registers; input and output tiles
// (64 streams x 32 f32 samples, or x 16 f64 samples) cross LDS so that global loads and stores are coalesced 16-byte
// accesses of row segments, as in the modulator's store_tile.  The reference's circular buffers (length order+1 and order,
// filters.ts:92-98) hold exactly the last `order` inputs and outputs besides the current one, which is what the registers hold.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <new>
#include <vector>

#include "fsk_host.h"

namespace fsk {
namespace {

constexpr int kIirMax = 8;                 // highest order
constexpr int kIirChunks = 8;              // 16-byte chunks per row per tile
constexpr int kIirStride = 65;             // 16-byte slots per chunk column (64 rows + 1 pad: conflict-free both ways)

struct IirCoef {
  double b[kIirMax + 1], a[kIirMax + 1];   // normalised (a[0] = 1)
  float bf[kIirMax + 1], af[kIirMax + 1];  // the same rounded once, for the fp32 path
  uint32_t nb, na;
};

template <typename Real>
__device__ __forceinline__ Real coef_b(const IirCoef &C, int i);
template <>
__device__ __forceinline__ double coef_b<double>(const IirCoef &C, int i) { return C.b[i]; }
template <>
__device__ __forceinline__ float coef_b<float>(const IirCoef &C, int i) { return C.bf[i]; }
template <typename Real>
__device__ __forceinline__ Real coef_a(const IirCoef &C, int i);
template <>
__device__ __forceinline__ double coef_a<double>(const IirCoef &C, int i) { return C.a[i]; }
template <>
__device__ __forceinline__ float coef_a<float>(const IirCoef &C, int i) { return C.af[i]; }

// one sample through the difference equation (filters.ts:47-76); xh[i] = x[n-1-i], yh[i] = y[n-1-i].
// NB = b.length, NA = a.length at compile time -- the kernel is instantiated for b.length = a.length = 1 .. 9, every filter
// FilterDesign produces: no test per tap (a lone wave pays ~35 cycles per branch: sixteen of them were 500 cycles per sample),
// only the history the order needs is moved -- or 0, 0: the lengths at run time (any other pair).
template <typename Real, int NB, int NA>
__device__ __forceinline__ Real iir_step(const IirCoef &C, Real (&xh)[kIirMax], Real (&yh)[kIirMax], Real x) {
  constexpr int H = NB ? (NB > NA ? NB : NA) - 1 : kIirMax;      // history entries in use
  Real out = (Real)0;
  out += coef_b<Real>(C, 0) * x;
#pragma unroll
  for (int i = 1; i <= kIirMax; i++)
    if (NB ? i < NB : (uint32_t)i < C.nb) out += coef_b<Real>(C, i) * xh[i - 1];
#pragma unroll
  for (int i = 1; i <= kIirMax; i++)
    if (NA ? i < NA : (uint32_t)i < C.na) out -= coef_a<Real>(C, i) * yh[i - 1];
#pragma unroll
  for (int i = H - 1; i > 0; i--) { xh[i] = xh[i - 1]; yh[i] = yh[i - 1]; }
  if (H > 0) { xh[0] = x; yh[0] = out; }
  return out;
}

// IO = float: processBuffer() (Float32Array in, Float32Array out); IO = double: process() sample by sample (numbers in, numbers out)
template <typename Real, typename IO, int NB, int NA>
// (in and out are NOT __restrict__: the header allows in == out, and tile t + 1 is loaded before tile t is stored -- different
// addresses, but an aliasing promise the language would let the compiler build on)
__global__ __launch_bounds__(64) void iir_kernel(IirCoef C, const IO *in, size_t n, size_t in_pitch,
                                                 IO *out, size_t out_pitch, int vec_ok, Real *__restrict__ hx,
                                                 Real *__restrict__ hy, uint32_t n_streams) {
  constexpr int VN = 16 / (int)sizeof(IO);               // samples per 16-byte chunk
  constexpr int TILE = kIirChunks * VN;                  // samples per row per tile
  typedef IO Vec __attribute__((ext_vector_type(VN)));
  __shared__ Vec stage[kIirChunks * kIirStride];
  const uint32_t lane = threadIdx.x;
  const uint32_t stream = blockIdx.x * 64u + lane;
  const bool valid = stream < n_streams;
  const uint32_t row = valid ? stream : n_streams - 1;
  Real xh[kIirMax], yh[kIirMax];
#pragma unroll
  for (int i = 0; i < kIirMax; i++) {
    xh[i] = hx[(size_t)i * n_streams + row];
    yh[i] = hy[(size_t)i * n_streams + row];
  }
  const uint32_t sub_row = lane / kIirChunks, chunk = lane % kIirChunks;
  constexpr int PASSES = kIirChunks;                     // 64 rows = PASSES x (64 / kIirChunks) rows of kIirChunks chunks
  // a tile's 64 x kIirChunks chunks, PASSES per lane: a row's 128 bytes contiguous.  Whole tiles of an aligned buffer take
  // the unconditional path (rows beyond the batch read the last row again; nothing of them is stored).
  auto load_tile = [&](size_t t0, Vec (&dst)[PASSES]) {
    const size_t c0 = t0 + (size_t)VN * chunk;
    // (the whole-tile test stays per pass: hoisted around the eight loads -- one branch, the loads back to back -- the kernel
    // was 25 % SLOWER, 3.86 against 3.02 ms at 16 384 streams x 48 000)
    const bool whole = vec_ok && t0 + TILE <= n;
#pragma unroll
    for (int i = 0; i < PASSES; i++) {
      const uint32_t r = blockIdx.x * 64u + (uint32_t)(64 / kIirChunks) * i + sub_row;
      const IO *src = in + (size_t)(r < n_streams ? r : n_streams - 1) * in_pitch + c0;
      if (whole) dst[i] = *reinterpret_cast<const Vec *>(src);
      else {
        Vec v = (Vec)(IO)0;
#pragma unroll
        for (int k = 0; k < VN; k++) if (c0 + k < n) v[k] = src[k];
        dst[i] = v;
      }
    }
  };
  auto store_tile = [&](size_t t0) {                             // stage_out -> global
    const size_t c0 = t0 + (size_t)VN * chunk;
    if (vec_ok && t0 + TILE <= n) {
#pragma unroll
      for (int i = 0; i < PASSES; i++) {
        const uint32_t lr = (uint32_t)(64 / kIirChunks) * i + sub_row;
        const uint32_t r = blockIdx.x * 64u + lr;
        const Vec v = stage[chunk * kIirStride + lr];
        if (r < n_streams) *reinterpret_cast<Vec *>(out + (size_t)r * out_pitch + c0) = v;
      }
    } else {
#pragma unroll 1
      for (int i = 0; i < PASSES; i++) {
        const uint32_t lr = (uint32_t)(64 / kIirChunks) * i + sub_row;
        const uint32_t r = blockIdx.x * 64u + lr;
        if (r >= n_streams || c0 >= n) continue;
        const Vec v = stage[chunk * kIirStride + lr];
        IO *dst = out + (size_t)r * out_pitch + c0;
#pragma unroll
        for (int k = 0; k < VN; k++) if (c0 + k < n) dst[k] = v[k];
      }
    }
  };
  // Per tile: the tile loaded an iteration ago goes into LDS and the next tile's loads are issued before this tile's rows are
  // worked: with one wave per compute unit nothing else hides a memory round trip (x 2.8 at 16 384 streams).  (Tried: the
  // previous tile's stores ahead of the next tile's loads through a second LDS tile, so that the wait for the loads has nothing
  // younger in front of it: 3.98 against 3.04 ms.)
  Vec pre[PASSES];
  if (n) load_tile(0, pre);
  for (size_t t0 = 0; t0 < n; t0 += TILE) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PASSES; i++) stage[chunk * kIirStride + (uint32_t)(64 / kIirChunks) * i + sub_row] = pre[i];
    if (t0 + TILE < n) load_tile(t0 + TILE, pre);
    __syncthreads();
    // ---- this lane's row, in time order
    const uint32_t len = (uint32_t)(n - t0 < (size_t)TILE ? n - t0 : (size_t)TILE);
#pragma unroll 1
    for (uint32_t c = 0; c < (uint32_t)kIirChunks; c++) {
      if (c * VN >= len) break;
      Vec v = stage[c * kIirStride + lane];
      if ((c + 1u) * VN <= len) {                                // a whole chunk: no test per sample
#pragma unroll
        for (int k = 0; k < VN; k++) v[k] = (IO)iir_step<Real, NB, NA>(C, xh, yh, (Real)v[k]);   // (the f32 store of processBuffer, filters.ts:84)
      } else {
#pragma unroll
        for (int k = 0; k < VN; k++) {
          if (c * VN + k < len) v[k] = (IO)iir_step<Real, NB, NA>(C, xh, yh, (Real)v[k]);
        }
      }
      stage[c * kIirStride + lane] = v;
    }
    __syncthreads();
    store_tile(t0);
  }
  if (valid) {
#pragma unroll
    for (int i = 0; i < kIirMax; i++) {
      hx[(size_t)i * n_streams + stream] = xh[i];
      hy[(size_t)i * n_streams + stream] = yh[i];
    }
  }
}

template <typename Real>
__global__ void iir_reset_kernel(Real *hx, Real *hy, uint32_t n_streams, int64_t stream) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)n_streams * kIirMax) return;
  if (stream >= 0 && (int64_t)(idx % n_streams) != stream) return;
  hx[idx] = (Real)0; hy[idx] = (Real)0;
}

}  // namespace
}  // namespace fsk

using namespace fsk;

struct fskhip_iir {
  int device = 0;
  int precision = 0;
  uint32_t S = 0;
  IirCoef C{};
  void *hx = nullptr, *hy = nullptr;   // [kIirMax][S] Real: x[n-1-i], y[n-1-i]
  hipStream_t stream = nullptr;
  void *d_in = nullptr, *d_out = nullptr; size_t d_in_cap = 0, d_out_cap = 0;   // staging of the _host entry points (bytes)
};

template <typename IO>
static int iir_process_device(fskhip_iir *f, const IO *d_in, size_t n, size_t in_pitch, IO *d_out, size_t out_pitch, void *hip_stream,
                              const char *who) {
  if (!f) return fail(FSKHIP_E_INVALID, "null filter");
  if (n == 0) return FSKHIP_OK;
  if (!d_in || !d_out) return fail(FSKHIP_E_INVALID, "%s: null buffer", who);
  if (in_pitch < n || out_pitch < n) return fail(FSKHIP_E_INVALID, "%s: pitch < n_per_stream", who);
  HIP_TRY(hipSetDevice(f->device));
  hipStream_t st = (hipStream_t)hip_stream;
  constexpr size_t VN = 16 / sizeof(IO);
  const int vec_ok = (in_pitch % VN == 0) && (out_pitch % VN == 0) && ((reinterpret_cast<uintptr_t>(d_in) & 15u) == 0) &&
                     ((reinterpret_cast<uintptr_t>(d_out) & 15u) == 0);
  dim3 g((f->S + 63u) / 64u), b(64);
#define FSK_IIR_LAUNCH(NBV, NAV)                                                                                              \
  do {                                                                                                                        \
    if (f->precision == FSKHIP_PRECISION_F64)                                                                                 \
      hipLaunchKernelGGL((iir_kernel<double, IO, NBV, NAV>), g, b, 0, st, f->C, d_in, n, in_pitch, d_out, out_pitch, vec_ok, \
                         (double *)f->hx, (double *)f->hy, f->S);                                                             \
    else                                                                                                                      \
      hipLaunchKernelGGL((iir_kernel<float, IO, NBV, NAV>), g, b, 0, st, f->C, d_in, n, in_pitch, d_out, out_pitch, vec_ok,  \
                         (float *)f->hx, (float *)f->hy, f->S);                                                               \
  } while (0)
  switch (f->C.nb == f->C.na ? f->C.nb : 0u) {
    case 1: FSK_IIR_LAUNCH(1, 1); break;
    case 2: FSK_IIR_LAUNCH(2, 2); break;
    case 3: FSK_IIR_LAUNCH(3, 3); break;
    case 4: FSK_IIR_LAUNCH(4, 4); break;
    case 5: FSK_IIR_LAUNCH(5, 5); break;
    case 6: FSK_IIR_LAUNCH(6, 6); break;
    case 7: FSK_IIR_LAUNCH(7, 7); break;
    case 8: FSK_IIR_LAUNCH(8, 8); break;
    case 9: FSK_IIR_LAUNCH(9, 9); break;
    default: FSK_IIR_LAUNCH(0, 0); break;
  }
#undef FSK_IIR_LAUNCH
  HIP_TRY(hipGetLastError());
  return FSKHIP_OK;
}

template <typename IO>
static int iir_process_host(fskhip_iir *f, const IO *in, size_t n, size_t in_pitch, IO *out, size_t out_pitch, const char *who) {
  if (!f) return fail(FSKHIP_E_INVALID, "null filter");
  if (n == 0) return FSKHIP_OK;
  if (!in || !out) return fail(FSKHIP_E_INVALID, "%s: null buffer", who);
  if (in_pitch < n || out_pitch < n) return fail(FSKHIP_E_INVALID, "%s: pitch < n_per_stream", who);
  HIP_TRY(hipSetDevice(f->device));
  constexpr size_t VN = 16 / sizeof(IO);
  const size_t dp = (n + VN - 1) / VN * VN, S = f->S;
  auto ensure = [&](void *&p, size_t &cap, size_t need) -> int {
    if (need <= cap) return FSKHIP_OK;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    hipError_t err = hipMalloc(&p, need);
    if (err != hipSuccess) return fail(FSKHIP_E_NOMEM, "hipMalloc(%zu): %s", need, hipGetErrorString(err));
    cap = need;
    return FSKHIP_OK;
  };
  int rc;
  if ((rc = ensure(f->d_in, f->d_in_cap, dp * S * sizeof(IO))) != FSKHIP_OK) return rc;
  if ((rc = ensure(f->d_out, f->d_out_cap, dp * S * sizeof(IO))) != FSKHIP_OK) return rc;
  HIP_TRY(hipMemcpy2DAsync(f->d_in, dp * sizeof(IO), in, in_pitch * sizeof(IO), n * sizeof(IO), S, hipMemcpyHostToDevice, f->stream));
  if ((rc = iir_process_device<IO>(f, (const IO *)f->d_in, n, dp, (IO *)f->d_out, dp, f->stream, who)) != FSKHIP_OK) return rc;
  HIP_TRY(hipMemcpy2DAsync(out, out_pitch * sizeof(IO), f->d_out, dp * sizeof(IO), n * sizeof(IO), S, hipMemcpyDeviceToHost, f->stream));
  HIP_TRY(hipStreamSynchronize(f->stream));
  return FSKHIP_OK;
}

extern "C" {

int fskhip_iir_destroy(fskhip_iir *f) {
  if (!f) return FSKHIP_OK;
  (void)hipSetDevice(f->device);
  (void)hipDeviceSynchronize();
  void *bufs[] = {f->hx, f->hy, f->d_in, f->d_out};
  for (void *b : bufs)
    if (b) (void)hipFree(b);
  if (f->stream) (void)hipStreamDestroy(f->stream);
  delete f;
  return FSKHIP_OK;
}

int fskhip_iir_create(int device, const double *b, uint32_t nb, const double *a, uint32_t na, uint32_t n_streams, int precision,
                      fskhip_iir **out) {
  if (!out || n_streams == 0) return fail(FSKHIP_E_INVALID, "fskhip_iir_create: null/zero argument");
  // the constructor's three errors, with the reference's messages (filters.ts:19-21)
  if (!b || nb == 0) return fail(FSKHIP_E_INVALID, "Feedforward coefficients (b) cannot be empty");
  if (!a || na == 0) return fail(FSKHIP_E_INVALID, "Feedback coefficients (a) cannot be empty");
  if (a[0] == 0.0) return fail(FSKHIP_E_INVALID, "First feedback coefficient (a[0]) cannot be zero");
  if (precision != FSKHIP_PRECISION_F32 && precision != FSKHIP_PRECISION_F64) return fail(FSKHIP_E_INVALID, "unknown precision %d", precision);
  if (nb > (uint32_t)kIirMax + 1u || na > (uint32_t)kIirMax + 1u)
    return fail(FSKHIP_E_UNSUPPORTED, "IIR order %u: the batched kernel keeps up to %d past inputs and outputs in registers", (nb > na ? nb : na) - 1u, kIirMax);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(FSKHIP_E_NO_DEVICE, "no HIP device available (the engine has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(FSKHIP_E_NO_DEVICE, "device %d out of range (%d devices)", device, ndev);
  if (hipSetDevice(device) != hipSuccess) return fail(FSKHIP_E_NO_DEVICE, "hipSetDevice(%d) failed", device);
  fskhip_iir *f = new (std::nothrow) fskhip_iir();
  if (!f) return fail(FSKHIP_E_NOMEM, "out of host memory");
  f->device = device; f->precision = precision; f->S = n_streams;
  f->C.nb = nb; f->C.na = na;
  for (uint32_t i = 0; i <= (uint32_t)kIirMax; i++) { f->C.b[i] = i < nb ? b[i] : 0.0; f->C.a[i] = i < na ? a[i] : 0.0; }
  if (f->C.a[0] != 1.0) {     // normalisation (filters.ts:30-39): b[i] /= a0, a[i] /= a0 for i >= 1, a[0] = 1
    const double a0 = f->C.a[0];
    for (uint32_t i = 0; i < nb; i++) f->C.b[i] /= a0;
    for (uint32_t i = 1; i < na; i++) f->C.a[i] /= a0;
    f->C.a[0] = 1.0;
  }
  for (uint32_t i = 0; i <= (uint32_t)kIirMax; i++) { f->C.bf[i] = (float)f->C.b[i]; f->C.af[i] = (float)f->C.a[i]; }
  const size_t rsz = precision == FSKHIP_PRECISION_F64 ? sizeof(double) : sizeof(float);
  const size_t hsz = rsz * (size_t)kIirMax * n_streams;
  hipError_t err = hipMalloc(&f->hx, hsz);
  if (err == hipSuccess) err = hipMalloc(&f->hy, hsz);
  if (err == hipSuccess) err = hipMemset(f->hx, 0, hsz);
  if (err == hipSuccess) err = hipMemset(f->hy, 0, hsz);
  if (err == hipSuccess) err = hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking);
  if (err == hipSuccess) err = hipDeviceSynchronize();
  if (err != hipSuccess) {
    fskhip_iir_destroy(f);
    return fail(err == hipErrorOutOfMemory ? FSKHIP_E_NOMEM : FSKHIP_E_HIP, "fskhip_iir_create: %s", hipGetErrorString(err));
  }
  *out = f;
  return FSKHIP_OK;
}

uint32_t fskhip_iir_streams(const fskhip_iir *f) { return f ? f->S : 0u; }

int fskhip_iir_get_coefficients(const fskhip_iir *f, double *b, uint32_t *nb, double *a, uint32_t *na) {
  if (!f || !b || !a || !nb || !na) return fail(FSKHIP_E_INVALID, "fskhip_iir_get_coefficients: null argument");
  for (uint32_t i = 0; i < f->C.nb; i++) b[i] = f->C.b[i];
  for (uint32_t i = 0; i < f->C.na; i++) a[i] = f->C.a[i];
  *nb = f->C.nb; *na = f->C.na;
  return FSKHIP_OK;
}

int fskhip_iir_process_device(fskhip_iir *f, const float *d_in, size_t n, size_t in_pitch, float *d_out, size_t out_pitch, void *hip_stream) {
  return iir_process_device<float>(f, d_in, n, in_pitch, d_out, out_pitch, hip_stream, "fskhip_iir_process_device");
}
int fskhip_iir_process_host(fskhip_iir *f, const float *in, size_t n, size_t in_pitch, float *out, size_t out_pitch) {
  return iir_process_host<float>(f, in, n, in_pitch, out, out_pitch, "fskhip_iir_process_host");
}
int fskhip_iir_process_f64_device(fskhip_iir *f, const double *d_in, size_t n, size_t in_pitch, double *d_out, size_t out_pitch, void *hip_stream) {
  return iir_process_device<double>(f, d_in, n, in_pitch, d_out, out_pitch, hip_stream, "fskhip_iir_process_f64_device");
}
int fskhip_iir_process_f64_host(fskhip_iir *f, const double *in, size_t n, size_t in_pitch, double *out, size_t out_pitch) {
  return iir_process_host<double>(f, in, n, in_pitch, out, out_pitch, "fskhip_iir_process_f64_host");
}

int fskhip_iir_reset(fskhip_iir *f, int64_t stream) {
  if (!f) return fail(FSKHIP_E_INVALID, "null filter");
  if (stream >= (int64_t)f->S) return fail(FSKHIP_E_INVALID, "stream out of range");
  HIP_TRY(hipSetDevice(f->device));
  HIP_TRY(hipDeviceSynchronize());
  const size_t total = (size_t)f->S * kIirMax;
  if (f->precision == FSKHIP_PRECISION_F64)
    hipLaunchKernelGGL(iir_reset_kernel<double>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, (double *)f->hx, (double *)f->hy, f->S, stream);
  else
    hipLaunchKernelGGL(iir_reset_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, (float *)f->hx, (float *)f->hy, f->S, stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  return FSKHIP_OK;
}

}  // extern "C"
