// fsk_fir.hip -- batched FIRFilter (src/dsp/filters.ts:112-167) and the windowed-sinc designs (243-314), SURVEY.md 8(f3).
//
// Unlike the IIR chain a FIR has no recurrence, so the batch is parallel over time as well as over streams: one
// workgroup filters a 1024-sample segment of one stream.  The segment (plus the n_taps-1 samples before it) is
// staged in LDS, the taps ride in SGPRs; each lane produces 4 consecutive outputs from a sliding window of two float4, so a single
// conflict-free ds_read_b128 feeds 16 multiply-adds.  Every output still accumulates its products in the
// reference's order (i = 0 .. n_taps-1, `output += c[i] * delay[i]`, doubles in the parity path).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <new>
#include <vector>

#include "fsk_host.h"

namespace fsk {
namespace {

constexpr int kFirBlock = 256;
constexpr int kFirPerLane = 4;
constexpr int kFirTile = kFirBlock * kFirPerLane;

template <typename Real>
__device__ __forceinline__ Real mac(Real acc, Real c, float x);
template <>
__device__ __forceinline__ double mac<double>(double acc, double c, float x) { return acc + c * (double)x; }  // two roundings (-ffp-contract=off)
template <>
__device__ __forceinline__ float mac<float>(float acc, float c, float x) { return __builtin_fmaf(c, x, acc); }

// hist holds the n_taps-1 inputs before sample 0 of this call, oldest first ([stream][n_taps-1])
template <typename Real>
__global__ __launch_bounds__(kFirBlock) void fir_kernel(const float *__restrict__ in, size_t n, size_t in_pitch,
                                                        float *__restrict__ out, size_t out_pitch, int vec_ok,
                                                        const float *__restrict__ hist, const Real *__restrict__ taps,
                                                        uint32_t n_taps, uint32_t tiles_per_row) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  const uint32_t pad = ((n_taps - 1 + 3) & ~3u) + 4;  // floats in front of the segment, multiple of 4
  float *xs = reinterpret_cast<float *>(lds_raw);                 // [pad + kFirTile]
  // the taps are wave-uniform: indexed by the (uniform) loop counter straight from the kernel-argument pointer
  // they come in through the scalar cache into SGPRs and cost no LDS or vector-memory traffic
  const Real *cs = taps;
  const uint32_t row = blockIdx.x / tiles_per_row;
  const size_t t0 = (size_t)(blockIdx.x % tiles_per_row) * kFirTile;
  const float *xrow = in + (size_t)row * in_pitch;
  const float *hrow = hist + (size_t)row * (n_taps - 1);
  for (uint32_t i = threadIdx.x; i < pad + kFirTile; i += kFirBlock) {
    const int64_t t = (int64_t)t0 + (int64_t)i - (int64_t)pad;
    float v = 0.0f;
    if (t >= 0) v = (size_t)t < n ? xrow[t] : 0.0f;
    else if (t >= -(int64_t)(n_taps - 1)) v = hrow[(int64_t)(n_taps - 1) + t];
    xs[i] = v;
  }
  __syncthreads();

  const uint32_t j = threadIdx.x;
  const float4 *x4 = reinterpret_cast<const float4 *>(xs);
  const uint32_t base4 = pad / 4 + j;  // float4 index of x[4j .. 4j+3]
  Real a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  float4 hi = x4[base4];
  for (uint32_t i0 = 0; i0 < n_taps; i0 += 4) {
    const float4 lo = x4[base4 - i0 / 4 - 1];
    // window w[0..7] = x[4j-i0-4 .. 4j-i0+3]; tap i0+d meets output k at w[4+k-d]
    const Real c0 = cs[i0];
    a0 = mac(a0, c0, hi.x); a1 = mac(a1, c0, hi.y); a2 = mac(a2, c0, hi.z); a3 = mac(a3, c0, hi.w);
    if (i0 + 1 < n_taps) {
      const Real c = cs[i0 + 1];
      a0 = mac(a0, c, lo.w); a1 = mac(a1, c, hi.x); a2 = mac(a2, c, hi.y); a3 = mac(a3, c, hi.z);
    }
    if (i0 + 2 < n_taps) {
      const Real c = cs[i0 + 2];
      a0 = mac(a0, c, lo.z); a1 = mac(a1, c, lo.w); a2 = mac(a2, c, hi.x); a3 = mac(a3, c, hi.y);
    }
    if (i0 + 3 < n_taps) {
      const Real c = cs[i0 + 3];
      a0 = mac(a0, c, lo.y); a1 = mac(a1, c, lo.z); a2 = mac(a2, c, lo.w); a3 = mac(a3, c, hi.x);
    }
    hi = lo;
  }
  const size_t t = t0 + 4u * j;
  if (t >= n) return;
  float *o = out + (size_t)row * out_pitch + t;
  if (vec_ok && t + 4 <= n) {
    *reinterpret_cast<float4 *>(o) = make_float4((float)a0, (float)a1, (float)a2, (float)a3);
  } else {
    o[0] = (float)a0;
    if (t + 1 < n) o[1] = (float)a1;
    if (t + 2 < n) o[2] = (float)a2;
    if (t + 3 < n) o[3] = (float)a3;
  }
}

// the delay line after the call: the last n_taps-1 inputs (old history where the call was shorter than that)
__global__ void fir_hist_kernel(const float *__restrict__ in, size_t n, size_t in_pitch, const float *__restrict__ hist_in,
                                float *__restrict__ hist_out, uint32_t n_taps, uint32_t n_streams) {
  const uint32_t h = n_taps - 1;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)n_streams * h) return;
  const uint32_t row = (uint32_t)(idx / h), k = (uint32_t)(idx % h);
  const int64_t t = (int64_t)n - (int64_t)h + (int64_t)k;
  hist_out[idx] = t >= 0 ? in[(size_t)row * in_pitch + (size_t)t] : hist_in[(size_t)row * h + (size_t)((int64_t)h + t)];
}

__global__ void fir_reset_kernel(float *hist, uint32_t h, uint32_t n_streams, int64_t stream) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)n_streams * h) return;
  if (stream >= 0 && (int64_t)(idx / h) != stream) return;
  hist[idx] = 0.0f;
}

int sinc_lowpass(double cutoff, double sampleRate, uint32_t n_taps, std::vector<double> &c) {  // filters.ts:243-265
  if (n_taps % 2 == 0) n_taps++;
  const double normalizedCutoff = cutoff / sampleRate;
  const double center = ((double)n_taps - 1) / 2;
  c.assign(n_taps, 0.0);
  for (uint32_t i = 0; i < n_taps; i++) {
    if ((double)i == center) {
      c[i] = 2 * normalizedCutoff;
    } else {
      const double x = M_PI * ((double)i - center);
      c[i] = std::sin(2 * normalizedCutoff * x) / x;
    }
    c[i] *= 0.54 - 0.46 * std::cos(2 * M_PI * (double)i / ((double)n_taps - 1));
  }
  return (int)n_taps;
}
// filters.ts:274-286: the loop bounds use the CALLER's numTaps while the array has sincLowpass's (odd) length;
// with an even numTaps `lowpass[center]` is a fractional index that touches no element
int sinc_highpass(double cutoff, double sampleRate, uint32_t n_taps, std::vector<double> &c) {
  const int n = sinc_lowpass(cutoff, sampleRate, n_taps, c);
  for (uint32_t i = 0; i < n_taps && i < (uint32_t)n; i++) c[i] = -c[i];
  if (n_taps % 2 == 1) c[(n_taps - 1) / 2] += 1;
  return n;
}

}  // namespace
}  // namespace fsk

using namespace fsk;

struct fskhip_fir {
  int device = 0;
  int precision = 0;
  uint32_t n_taps = 0, S = 0;
  double *d_taps = nullptr;   // coefficients as given (fp64 path)
  float *d_taps32 = nullptr;  // rounded once on the host (fp32 path)
  float *hist[2] = {nullptr, nullptr};  // ping-pong: [cur] is read by the next call
  int cur = 0;
  hipStream_t stream = nullptr;
  float *d_in = nullptr, *d_out = nullptr; size_t d_in_cap = 0, d_out_cap = 0;
};

extern "C" {

int fskhip_sinc_lowpass(double cutoff, double sampleRate, uint32_t n_taps, double *taps) {
  if (!taps || n_taps == 0) return fail(FSKHIP_E_INVALID, "fskhip_sinc_lowpass: bad argument");
  std::vector<double> c;
  const int n = sinc_lowpass(cutoff, sampleRate, n_taps, c);
  for (int i = 0; i < n; i++) taps[i] = c[i];
  return n;
}
int fskhip_sinc_highpass(double cutoff, double sampleRate, uint32_t n_taps, double *taps) {
  if (!taps || n_taps == 0) return fail(FSKHIP_E_INVALID, "fskhip_sinc_highpass: bad argument");
  std::vector<double> c;
  const int n = sinc_highpass(cutoff, sampleRate, n_taps, c);
  for (int i = 0; i < n; i++) taps[i] = c[i];
  return n;
}
int fskhip_sinc_bandpass(double center, double bandwidth, double sampleRate, uint32_t n_taps, double *taps) {  // filters.ts:296-314
  if (!taps || n_taps == 0) return fail(FSKHIP_E_INVALID, "fskhip_sinc_bandpass: bad argument");
  // (an even numTaps works in the reference too: its sincHighpass / sincLowpass return numTaps + 1 taps then, of which
  // the convolution below only reads the first numTaps)
  std::vector<double> hp, lp;
  sinc_highpass(center - bandwidth / 2, sampleRate, n_taps, hp);
  sinc_lowpass(center + bandwidth / 2, sampleRate, n_taps, lp);
  for (uint32_t i = 0; i < n_taps; i++) taps[i] = 0;
  for (uint32_t i = 0; i < n_taps; i++)
    for (uint32_t j = 0; j < n_taps; j++)
      if (i + j < n_taps) taps[i + j] += hp[i] * lp[j];
  return (int)n_taps;
}

uint32_t fskhip_fir_streams(const fskhip_fir *f) { return f ? f->S : 0u; }

int fskhip_fir_destroy(fskhip_fir *f) {
  if (!f) return FSKHIP_OK;
  (void)hipSetDevice(f->device);
  (void)hipDeviceSynchronize();
  void *bufs[] = {f->d_taps, f->d_taps32, f->hist[0], f->hist[1], f->d_in, f->d_out};
  for (void *b : bufs)
    if (b) (void)hipFree(b);
  if (f->stream) (void)hipStreamDestroy(f->stream);
  delete f;
  return FSKHIP_OK;
}

int fskhip_fir_create(int device, const double *taps, uint32_t n_taps, uint32_t n_streams, int precision, fskhip_fir **out) {
  if (!taps || !out || n_taps == 0 || n_streams == 0) return fail(FSKHIP_E_INVALID, "fskhip_fir_create: null/zero argument");
  if (precision != FSKHIP_PRECISION_F32 && precision != FSKHIP_PRECISION_F64) return fail(FSKHIP_E_INVALID, "unknown precision %d", precision);
  if (n_taps > 4096) return fail(FSKHIP_E_UNSUPPORTED, "%u taps do not fit the LDS tile (max 4096)", n_taps);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(FSKHIP_E_NO_DEVICE, "no HIP device available (the engine has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(FSKHIP_E_NO_DEVICE, "device %d out of range (%d devices)", device, ndev);
  if (hipSetDevice(device) != hipSuccess) return fail(FSKHIP_E_NO_DEVICE, "hipSetDevice(%d) failed", device);
  fskhip_fir *f = new (std::nothrow) fskhip_fir();
  if (!f) return fail(FSKHIP_E_NOMEM, "out of host memory");
  f->device = device; f->precision = precision; f->n_taps = n_taps; f->S = n_streams;
  const size_t hsz = sizeof(float) * (size_t)n_streams * (n_taps > 1 ? n_taps - 1 : 1);
  hipError_t err = hipMalloc((void **)&f->d_taps, sizeof(double) * n_taps);
  if (err == hipSuccess) err = hipMalloc((void **)&f->hist[0], hsz);
  if (err == hipSuccess) err = hipMalloc((void **)&f->hist[1], hsz);
  if (err == hipSuccess) err = hipMalloc((void **)&f->d_taps32, sizeof(float) * n_taps);
  if (err == hipSuccess) err = hipMemcpy(f->d_taps, taps, sizeof(double) * n_taps, hipMemcpyHostToDevice);
  if (err == hipSuccess) {
    std::vector<float> t32(n_taps);
    for (uint32_t i = 0; i < n_taps; i++) t32[i] = (float)taps[i];
    err = hipMemcpy(f->d_taps32, t32.data(), sizeof(float) * n_taps, hipMemcpyHostToDevice);
  }
  if (err == hipSuccess) err = hipMemset(f->hist[0], 0, hsz);
  if (err == hipSuccess) err = hipMemset(f->hist[1], 0, hsz);
  if (err == hipSuccess) err = hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking);
  if (err == hipSuccess) err = hipDeviceSynchronize();
  if (err != hipSuccess) {
    fskhip_fir_destroy(f);
    return fail(err == hipErrorOutOfMemory ? FSKHIP_E_NOMEM : FSKHIP_E_HIP, "fskhip_fir_create: %s", hipGetErrorString(err));
  }
  *out = f;
  return FSKHIP_OK;
}

int fskhip_fir_process_device(fskhip_fir *f, const float *d_in, size_t n, size_t in_pitch, float *d_out, size_t out_pitch,
                              void *hip_stream) {
  if (!f) return fail(FSKHIP_E_INVALID, "null filter");
  if (n == 0) return FSKHIP_OK;
  if (!d_in || !d_out) return fail(FSKHIP_E_INVALID, "fskhip_fir_process_device: null buffer");
  if (in_pitch < n || out_pitch < n) return fail(FSKHIP_E_INVALID, "pitch < n_per_stream");
  HIP_TRY(hipSetDevice(f->device));
  hipStream_t st = (hipStream_t)hip_stream;
  const uint32_t tiles = (uint32_t)((n + kFirTile - 1) / kFirTile);
  if ((uint64_t)tiles * f->S > 0x7FFFFFFFull) return fail(FSKHIP_E_INVALID, "too many tiles for one launch");
  const uint32_t pad = ((f->n_taps - 1 + 3) & ~3u) + 4;
  const size_t lds = sizeof(float) * (pad + kFirTile);
  const int vec_ok = (out_pitch % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_out) & 15u) == 0);
  dim3 g(tiles * f->S), b(kFirBlock);
  if (f->precision == FSKHIP_PRECISION_F64)
    hipLaunchKernelGGL(fir_kernel<double>, g, b, lds, st, d_in, n, in_pitch, d_out, out_pitch, vec_ok, f->hist[f->cur],
                       f->d_taps, f->n_taps, tiles);
  else
    hipLaunchKernelGGL(fir_kernel<float>, g, b, lds, st, d_in, n, in_pitch, d_out, out_pitch, vec_ok, f->hist[f->cur],
                       f->d_taps32, f->n_taps, tiles);
  HIP_TRY(hipGetLastError());
  if (f->n_taps > 1) {
    const size_t total = (size_t)f->S * (f->n_taps - 1);
    hipLaunchKernelGGL(fir_hist_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_in, n, in_pitch,
                       f->hist[f->cur], f->hist[f->cur ^ 1], f->n_taps, f->S);
    HIP_TRY(hipGetLastError());
    f->cur ^= 1;
  }
  return FSKHIP_OK;
}

int fskhip_fir_process_host(fskhip_fir *f, const float *in, size_t n, size_t in_pitch, float *out, size_t out_pitch) {
  if (!f) return fail(FSKHIP_E_INVALID, "null filter");
  if (n == 0) return FSKHIP_OK;
  if (!in || !out) return fail(FSKHIP_E_INVALID, "fskhip_fir_process_host: null buffer");
  if (in_pitch < n || out_pitch < n) return fail(FSKHIP_E_INVALID, "pitch < n_per_stream");
  HIP_TRY(hipSetDevice(f->device));
  const size_t dp = (n + 3) & ~(size_t)3, S = f->S;
  auto ensure = [&](float *&p, size_t &cap, size_t need) -> int {
    if (need <= cap) return FSKHIP_OK;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    hipError_t err = hipMalloc((void **)&p, need * sizeof(float));
    if (err != hipSuccess) return fail(FSKHIP_E_NOMEM, "hipMalloc(%zu): %s", need * sizeof(float), hipGetErrorString(err));
    cap = need;
    return FSKHIP_OK;
  };
  int rc;
  if ((rc = ensure(f->d_in, f->d_in_cap, dp * S)) != FSKHIP_OK) return rc;
  if ((rc = ensure(f->d_out, f->d_out_cap, dp * S)) != FSKHIP_OK) return rc;
  HIP_TRY(hipMemcpy2DAsync(f->d_in, dp * sizeof(float), in, in_pitch * sizeof(float), n * sizeof(float), S,
                           hipMemcpyHostToDevice, f->stream));
  if ((rc = fskhip_fir_process_device(f, f->d_in, n, dp, f->d_out, dp, f->stream)) != FSKHIP_OK) return rc;
  HIP_TRY(hipMemcpy2DAsync(out, out_pitch * sizeof(float), f->d_out, dp * sizeof(float), n * sizeof(float), S,
                           hipMemcpyDeviceToHost, f->stream));
  HIP_TRY(hipStreamSynchronize(f->stream));
  return FSKHIP_OK;
}

int fskhip_fir_reset(fskhip_fir *f, int64_t stream) {
  if (!f) return fail(FSKHIP_E_INVALID, "null filter");
  if (stream >= (int64_t)f->S) return fail(FSKHIP_E_INVALID, "stream out of range");
  HIP_TRY(hipSetDevice(f->device));
  HIP_TRY(hipDeviceSynchronize());
  if (f->n_taps > 1) {
    const size_t total = (size_t)f->S * (f->n_taps - 1);
    hipLaunchKernelGGL(fir_reset_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, f->hist[f->cur],
                       f->n_taps - 1, f->S, stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
  }
  return FSKHIP_OK;
}

}  // extern "C"
