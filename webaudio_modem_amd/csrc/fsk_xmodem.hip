// fsk_xmodem.hip -- CRC-16-CCITT and XModem packet build / validation over batches of byte rows.
//
// What it restates: src/utils/crc16.ts:21-38 (CRC16.calculate), src/transports/xmodem/packet.ts:21-54
// (createData + serialize) and the receive checks of src/transports/xmodem/xmodem.ts:233-320 applied to the bytes a
// demodulate call returned.  One lane per row, bytes walked in order (the grammar is sequential per stream), the
// 256-entry CRC table built in LDS by the workgroup.  Byte/integer work: HBM-bound, no MFMA.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "fsk_host.h"

namespace fsk {
namespace {

constexpr uint32_t kSOH = 0x01, kEOT = 0x04;  // types.ts:29-34
constexpr int kBlock = 256;

// table[i] = CRC of the single byte i from a zero register: the 8 shift/xor steps of crc16.ts:25-33
__device__ __forceinline__ void build_crc_table(uint32_t *table) {
  for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) {
    uint32_t c = i << 8;
#pragma unroll
    for (int k = 0; k < 8; k++) c = (c & 0x8000u) ? ((c << 1) ^ 0x1021u) : (c << 1);
    table[i] = c & 0xFFFFu;
  }
  __syncthreads();
}
// crc ^= byte << 8, then 8 steps == one table step on the high byte
__device__ __forceinline__ uint32_t crc_step(const uint32_t *table, uint32_t crc, uint32_t byte) {
  return ((crc << 8) & 0xFFFFu) ^ table[((crc >> 8) ^ byte) & 0xFFu];
}

// Sequential byte reader over one row: a dword at a time when the row is 4-byte aligned.
template <bool ALIGNED>
struct RowReader {
  const uint8_t *row;
  uint32_t word;
  __device__ __forceinline__ explicit RowReader(const uint8_t *r) : row(r), word(0) {}
  __device__ __forceinline__ uint32_t get(uint32_t pos) {
    if (ALIGNED) {
      if ((pos & 3u) == 0) word = *(const uint32_t *)(row + pos);
      const uint32_t b = word & 0xFFu;
      word >>= 8;
      return b;
    }
    return row[pos];
  }
};

template <bool ALIGNED>
__global__ __launch_bounds__(kBlock) void crc16_kernel(const uint8_t *data, size_t pitch, const uint32_t *lens,
                                                        uint32_t n_rows, uint16_t *out) {
  __shared__ uint32_t table[256];
  build_crc_table(table);
  const uint32_t r = blockIdx.x * kBlock + threadIdx.x;
  if (r >= n_rows) return;
  RowReader<ALIGNED> rd(data + (size_t)r * pitch);
  const uint32_t n = lens[r];
  uint32_t crc = 0xFFFFu;  // crc16.ts:13
  for (uint32_t i = 0; i < n; i++) crc = crc_step(table, crc, rd.get(i));
  out[r] = (uint16_t)crc;
}

template <bool ALIGNED>
__global__ __launch_bounds__(kBlock) void serialize_kernel(const uint8_t *payloads, size_t payload_pitch,
                                                            const uint32_t *lens, const uint32_t *seqs, uint32_t n_rows,
                                                            uint8_t *out, size_t out_pitch, uint32_t *out_lens) {
  __shared__ uint32_t table[256];
  build_crc_table(table);
  const uint32_t r = blockIdx.x * kBlock + threadIdx.x;
  if (r >= n_rows) return;
  const uint32_t len = lens[r], seq = seqs[r];
  // createData's checks (packet.ts:22-27); also refuse rows that do not fit the slab
  if (seq < 1 || seq > 255 || len > 255 || (size_t)len + 6 > out_pitch) {
    out_lens[r] = 0;
    return;
  }
  RowReader<ALIGNED> rd(payloads + (size_t)r * payload_pitch);
  uint8_t *o = out + (size_t)r * out_pitch;
  o[0] = (uint8_t)kSOH;
  o[1] = (uint8_t)seq;
  o[2] = (uint8_t)(~seq & 0xFFu);
  o[3] = (uint8_t)len;
  uint32_t crc = 0xFFFFu;
  for (uint32_t i = 0; i < len; i++) {
    const uint32_t b = rd.get(i);
    o[4 + i] = (uint8_t)b;
    crc = crc_step(table, crc, b);
  }
  o[4 + len] = (uint8_t)(crc >> 8);
  o[5 + len] = (uint8_t)(crc & 0xFFu);
  out_lens[r] = len + 6;
}

// per-byte state machine of the receive grammar; every lane walks its own burst front to back
enum : uint32_t { ST_IDLE, ST_SEQ, ST_NSEQ, ST_LEN, ST_PAYLOAD, ST_CRC_HI, ST_CRC_LO, ST_DONE };

struct Scan {
  uint32_t state, status, expected;
  uint32_t seq, nseq, len, k, crc, rx, start;
  uint32_t packets, dropped, consumed, data_len;
  int32_t err_seq, err_len, crc_rx, crc_calc;
  bool accept;
  uint32_t word;  // payload bytes on their way to data[]: stored a dword at a time where the row allows it

  __device__ __forceinline__ void init(uint32_t expected_seq) {
    state = ST_IDLE; status = FSKHIP_XM_NEED_MORE; expected = expected_seq;
    seq = nseq = len = k = crc = rx = start = 0;
    packets = dropped = consumed = data_len = 0;
    err_seq = err_len = crc_rx = crc_calc = -1;
    accept = false;
    word = 0;
  }

  // assembleData (xmodem.ts:322-333): byte `off` of the stream's assembled payload.  Tentative until the packet's CRC
  // has matched -- only data[0 .. data_len) is meaningful afterwards.  With a 4-byte aligned row the bytes are merged
  // into dwords (one store per 4 bytes); a partial dword is flushed bytewise when its packet ends.
  template <bool DW>
  __device__ __forceinline__ void put(uint8_t *drow, size_t data_pitch, uint32_t off, uint32_t b, bool last) {
    if (!drow || (size_t)off >= data_pitch) return;
    if (!DW) { drow[off] = (uint8_t)b; return; }
    const uint32_t sh = (off & 3u) * 8u;
    word = (word & ~(0xFFu << sh)) | (b << sh);
    if ((off & 3u) == 3u && (size_t)off < data_pitch) {
      // full dword: bytes before this packet's first byte inside it were written by an earlier flush and are in `word`
      *reinterpret_cast<uint32_t *>(drow + (off & ~3u)) = word;
    } else if (last) {
      for (uint32_t q = off & ~3u; q <= off; q++) drow[q] = (uint8_t)(word >> ((q & 3u) * 8u));
    }
  }

  template <bool DW>
  __device__ __forceinline__ void byte(const uint32_t *table, uint32_t b, uint32_t pos, uint8_t *drow, size_t data_pitch) {
    switch (state) {
      case ST_IDLE:  // xmodem.ts:238-252
        if (b == kEOT) {
          status = FSKHIP_XM_EOT;
          state = ST_DONE;
        } else if (b == kSOH) {
          start = pos;
          state = ST_SEQ;
        }
        consumed = pos + 1;
        break;
      case ST_SEQ:
        seq = b;
        state = ST_NSEQ;
        break;
      case ST_NSEQ:
        nseq = b;
        state = ST_LEN;
        break;
      case ST_LEN: {  // xmodem.ts:266-274, 278, 309, 315
        len = b;
        const uint32_t prev = expected == 1 ? 255u : expected - 1;
        if (seq + nseq != 255u) {
          status = FSKHIP_XM_INVALID_SEQUENCE;
        } else if (seq == expected) {
          accept = true;
        } else if (seq == prev) {
          accept = false;
        } else {
          status = FSKHIP_XM_UNEXPECTED_SEQUENCE;
        }
        if (status != FSKHIP_XM_NEED_MORE) {
          err_seq = (int32_t)seq;
          err_len = (int32_t)len;
          dropped++;
          consumed = pos + 1;
          state = ST_DONE;
        } else {
          k = 0;
          crc = 0xFFFFu;
          state = len ? ST_PAYLOAD : ST_CRC_HI;
        }
        break;
      }
      case ST_PAYLOAD:
        if (accept) {
          put<DW>(drow, data_pitch, data_len + k, b, k + 1 == len);
          crc = crc_step(table, crc, b);
        }
        if (++k == len) state = ST_CRC_HI;
        break;
      case ST_CRC_HI:
        rx = b << 8;
        state = ST_CRC_LO;
        break;
      case ST_CRC_LO:
        rx |= b;
        consumed = pos + 1;
        state = ST_IDLE;
        if (accept) {
          packets++;        // statistics.packetsReceived: counted once the payload is in, before the CRC check (xmodem.ts:280)
          if (rx != crc) {  // xmodem.ts:287-291
            status = FSKHIP_XM_INVALID_CRC;
            err_seq = (int32_t)seq;
            err_len = (int32_t)len;
            crc_rx = (int32_t)rx;
            crc_calc = (int32_t)crc;
            dropped++;
            state = ST_DONE;
          } else {  // xmodem.ts:293-303
            data_len += len;
            expected = (expected % 255u) + 1;
          }
        } else {
          dropped++;  // duplicate: consumed and ignored (xmodem.ts:309-314)
        }
        break;
      default:
        break;
    }
  }

  __device__ __forceinline__ void finish(fskhip_xmodem_result *out) {
    if (state != ST_IDLE && state != ST_DONE) {  // ran out of bytes inside a packet (the reference's wait times out)
      status = FSKHIP_XM_TRUNCATED;
      // what waitForBytes has taken out of the receive buffer by then (xmodem.ts:475-499): SOH, and the three header
      // bytes once they were all there -- pinned to the real XModemTransport by tests/golden/manifest_next.json
      consumed = state >= ST_PAYLOAD ? start + 4u : start + 1u;
      if (state >= ST_PAYLOAD) {
        err_seq = (int32_t)seq;
        err_len = (int32_t)len;
      }
    }
    fskhip_xmodem_result r;
    r.status = status;
    r.expected_after = expected;
    r.packets = packets;
    r.dropped = dropped;
    r.consumed = consumed;
    r.data_len = data_len;
    r.err_seq = err_seq;
    r.err_len = err_len;
    r.crc_rx = crc_rx;
    r.crc_calc = crc_calc;
    *out = r;
  }
};

// any layout: every lane walks its own row
template <bool ALIGNED>
__global__ __launch_bounds__(kBlock) void scan_kernel(const uint8_t *bytes, size_t pitch, const uint32_t *counts,
                                                       const uint32_t *expected_in, uint32_t n_streams, uint8_t *data,
                                                       size_t data_pitch, fskhip_xmodem_result *results) {
  __shared__ uint32_t table[256];
  build_crc_table(table);
  const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
  if (s >= n_streams) return;
  RowReader<ALIGNED> rd(bytes + (size_t)s * pitch);
  uint8_t *drow = data ? data + (size_t)s * data_pitch : nullptr;
  const uint32_t n = counts[s];
  Scan sc;
  sc.init(expected_in[s]);
  for (uint32_t pos = 0; pos < n && sc.state != ST_DONE; pos++) sc.byte<false>(table, rd.get(pos), pos, drow, data_pitch);
  sc.finish(&results[s]);
}

// 16-byte aligned rows (the demodulator's own output slab qualifies when its pitch is a multiple of 16): one wave per
// 64 rows walks them in 64-byte tiles staged through LDS exactly like the demodulator's sample tiles -- four coalesced
// 16-B/lane loads of 16 rows x 64 B, chunk-major with a one-slot pad, then every lane reads its own 64 bytes -- instead
// of 64 lanes each chasing its own cache line one dword at a time.
__global__ __launch_bounds__(64) void scan_tiled_kernel(const uint8_t *bytes, size_t pitch, const uint32_t *counts,
                                                         const uint32_t *expected_in, uint32_t n_streams, uint8_t *data,
                                                         size_t data_pitch, int data_dw, fskhip_xmodem_result *results) {
  __shared__ uint32_t table[256];
  __shared__ uint4 stage[4 * 65];
  build_crc_table(table);
  const uint32_t lane = threadIdx.x;
  const uint32_t s = blockIdx.x * 64u + lane;
  const bool valid = s < n_streams;
  uint8_t *drow = (data && valid) ? data + (size_t)s * data_pitch : nullptr;
  const uint32_t n = valid ? counts[s] : 0u;
  uint32_t n_max = n;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const uint32_t t = __shfl_xor(n_max, o, 64); n_max = t > n_max ? t : n_max; }
  Scan sc;
  sc.init(valid ? expected_in[s] : 1u);
  const uint32_t sub_row = lane >> 2, chunk = lane & 3;
  for (uint32_t t0 = 0; t0 < n_max; t0 += 64u) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t r = blockIdx.x * 64u + 16u * i + sub_row;
      const size_t off = (size_t)t0 + 16u * chunk;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (r < n_streams && off + 16u <= pitch) v = *reinterpret_cast<const uint4 *>(bytes + (size_t)r * pitch + off);
      stage[chunk * 65u + 16u * i + sub_row] = v;
    }
    __syncthreads();
    if (t0 < n && sc.state != ST_DONE) {
#pragma unroll 1
      for (uint32_t c = 0; c < 4; c++) {
        const uint4 v = stage[c * 65u + lane];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 16; q++) {
          const uint32_t pos = t0 + 16u * c + (uint32_t)q;
          if (pos < n && sc.state != ST_DONE) {
            const uint32_t b = (w[q >> 2] >> ((q & 3) * 8)) & 0xFFu;
            if (data_dw) sc.byte<true>(table, b, pos, drow, data_pitch);
            else sc.byte<false>(table, b, pos, drow, data_pitch);
          }
        }
      }
    }
  }
  if (valid) sc.finish(&results[s]);
}

bool aligned4(const void *p, size_t pitch) { return ((uintptr_t)p & 3u) == 0 && (pitch & 3u) == 0; }

// scratch device buffers of one _host call
struct DevBufs {
  std::vector<void *> ptrs;
  ~DevBufs() {
    for (void *p : ptrs)
      if (p) (void)hipFree(p);
  }
  template <typename T>
  int alloc(T *&p, size_t n) {
    void *q = nullptr;
    hipError_t err = hipMalloc(&q, (n ? n : 1) * sizeof(T));
    if (err != hipSuccess) return fail(FSKHIP_E_NOMEM, "hipMalloc(%zu): %s", n * sizeof(T), hipGetErrorString(err));
    ptrs.push_back(q);
    p = (T *)q;
    return FSKHIP_OK;
  }
};

int select_device(int device) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(FSKHIP_E_NO_DEVICE, "no HIP device available (the engine has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(FSKHIP_E_NO_DEVICE, "device %d out of range (%d devices)", device, ndev);
  if (hipSetDevice(device) != hipSuccess) return fail(FSKHIP_E_NO_DEVICE, "hipSetDevice(%d) failed", device);
  return FSKHIP_OK;
}

}  // namespace
}  // namespace fsk

using namespace fsk;

extern "C" {

int fskhip_crc16_device(const uint8_t *d_data, size_t pitch, const uint32_t *d_lens, uint32_t n_rows, uint16_t *d_crc,
                        void *hip_stream) {
  if (n_rows == 0) return FSKHIP_OK;
  if (!d_lens || !d_crc) return fail(FSKHIP_E_INVALID, "fskhip_crc16_device: null buffer");
  hipStream_t st = (hipStream_t)hip_stream;
  dim3 g((n_rows + kBlock - 1) / kBlock), b(kBlock);
  if (aligned4(d_data, pitch)) hipLaunchKernelGGL(crc16_kernel<true>, g, b, 0, st, d_data, pitch, d_lens, n_rows, d_crc);
  else hipLaunchKernelGGL(crc16_kernel<false>, g, b, 0, st, d_data, pitch, d_lens, n_rows, d_crc);
  HIP_TRY(hipGetLastError());
  return FSKHIP_OK;
}

int fskhip_crc16_host(int device, const uint8_t *data, size_t pitch, const uint32_t *lens, uint32_t n_rows,
                      uint16_t *crc) {
  if (n_rows == 0) return FSKHIP_OK;
  if (!lens || !crc) return fail(FSKHIP_E_INVALID, "fskhip_crc16_host: null buffer");
  for (uint32_t r = 0; r < n_rows; r++)
    if (lens[r] > pitch) return fail(FSKHIP_E_INVALID, "lens[%u] = %u exceeds pitch %zu", r, lens[r], pitch);
  int rc = select_device(device);
  if (rc != FSKHIP_OK) return rc;
  DevBufs B;
  uint8_t *d_data = nullptr; uint32_t *d_lens = nullptr; uint16_t *d_crc = nullptr;
  const size_t dp = (pitch + 3) & ~(size_t)3;
  if ((rc = B.alloc(d_data, dp * n_rows)) || (rc = B.alloc(d_lens, n_rows)) || (rc = B.alloc(d_crc, n_rows))) return rc;
  if (pitch) HIP_TRY(hipMemcpy2D(d_data, dp ? dp : 4, data, pitch, pitch, n_rows, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_lens, lens, sizeof(uint32_t) * n_rows, hipMemcpyHostToDevice));
  if ((rc = fskhip_crc16_device(d_data, dp, d_lens, n_rows, d_crc, nullptr)) != FSKHIP_OK) return rc;
  HIP_TRY(hipMemcpy(crc, d_crc, sizeof(uint16_t) * n_rows, hipMemcpyDeviceToHost));
  return FSKHIP_OK;
}

int fskhip_xmodem_serialize_device(const uint8_t *d_payloads, size_t payload_pitch, const uint32_t *d_lens,
                                   const uint32_t *d_seqs, uint32_t n_rows, uint8_t *d_out, size_t out_pitch,
                                   uint32_t *d_out_lens, void *hip_stream) {
  if (n_rows == 0) return FSKHIP_OK;
  if (!d_lens || !d_seqs || !d_out || !d_out_lens) return fail(FSKHIP_E_INVALID, "fskhip_xmodem_serialize_device: null buffer");
  hipStream_t st = (hipStream_t)hip_stream;
  dim3 g((n_rows + kBlock - 1) / kBlock), b(kBlock);
  if (aligned4(d_payloads, payload_pitch))
    hipLaunchKernelGGL(serialize_kernel<true>, g, b, 0, st, d_payloads, payload_pitch, d_lens, d_seqs, n_rows, d_out,
                       out_pitch, d_out_lens);
  else
    hipLaunchKernelGGL(serialize_kernel<false>, g, b, 0, st, d_payloads, payload_pitch, d_lens, d_seqs, n_rows, d_out,
                       out_pitch, d_out_lens);
  HIP_TRY(hipGetLastError());
  return FSKHIP_OK;
}

int fskhip_xmodem_serialize_host(int device, const uint8_t *payloads, size_t payload_pitch, const uint32_t *lens,
                                 const uint32_t *seqs, uint32_t n_rows, uint8_t *out, size_t out_pitch,
                                 uint32_t *out_lens) {
  if (n_rows == 0) return FSKHIP_OK;
  if (!lens || !seqs || !out || !out_lens) return fail(FSKHIP_E_INVALID, "fskhip_xmodem_serialize_host: null buffer");
  for (uint32_t r = 0; r < n_rows; r++) {  // createData's throws, same texts (packet.ts:22-27)
    if (seqs[r] < 1 || seqs[r] > 255) return fail(FSKHIP_E_INVALID, "Invalid sequence: %u. Must be 1-255.", seqs[r]);
    if (lens[r] > 255) return fail(FSKHIP_E_INVALID, "Payload too large: %u. Max 255 bytes.", lens[r]);
    if (lens[r] > payload_pitch) return fail(FSKHIP_E_INVALID, "lens[%u] = %u exceeds payload_pitch %zu", r, lens[r], payload_pitch);
    if ((size_t)lens[r] + 6 > out_pitch) return fail(FSKHIP_E_OVERFLOW, "row %u needs %u bytes, slab holds %zu", r, lens[r] + 6, out_pitch);
  }
  int rc = select_device(device);
  if (rc != FSKHIP_OK) return rc;
  DevBufs B;
  uint8_t *d_p = nullptr, *d_o = nullptr; uint32_t *d_l = nullptr, *d_s = nullptr, *d_ol = nullptr;
  const size_t dp = (payload_pitch + 3) & ~(size_t)3;
  if ((rc = B.alloc(d_p, dp * n_rows)) || (rc = B.alloc(d_o, out_pitch * n_rows)) || (rc = B.alloc(d_l, n_rows)) ||
      (rc = B.alloc(d_s, n_rows)) || (rc = B.alloc(d_ol, n_rows)))
    return rc;
  if (payload_pitch) HIP_TRY(hipMemcpy2D(d_p, dp, payloads, payload_pitch, payload_pitch, n_rows, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_l, lens, sizeof(uint32_t) * n_rows, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_s, seqs, sizeof(uint32_t) * n_rows, hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(d_o, 0, out_pitch * n_rows));
  if ((rc = fskhip_xmodem_serialize_device(d_p, dp, d_l, d_s, n_rows, d_o, out_pitch, d_ol, nullptr)) != FSKHIP_OK) return rc;
  HIP_TRY(hipMemcpy(out, d_o, out_pitch * n_rows, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out_lens, d_ol, sizeof(uint32_t) * n_rows, hipMemcpyDeviceToHost));
  return FSKHIP_OK;
}

int fskhip_xmodem_scan_device(const uint8_t *d_bytes, size_t pitch, const uint32_t *d_counts, const uint32_t *d_expected,
                              uint32_t n_streams, uint8_t *d_data, size_t data_pitch, fskhip_xmodem_result *d_results,
                              void *hip_stream) {
  if (n_streams == 0) return FSKHIP_OK;
  if (!d_counts || !d_expected || !d_results) return fail(FSKHIP_E_INVALID, "fskhip_xmodem_scan_device: null buffer");
  hipStream_t st = (hipStream_t)hip_stream;
  dim3 g((n_streams + kBlock - 1) / kBlock), b(kBlock);
  if (((uintptr_t)d_bytes & 15u) == 0 && (pitch & 15u) == 0 && pitch >= 16) {
    const int data_dw = d_data && ((uintptr_t)d_data & 3u) == 0 && (data_pitch & 3u) == 0;
    hipLaunchKernelGGL(scan_tiled_kernel, dim3((n_streams + 63u) / 64u), dim3(64), 0, st, d_bytes, pitch, d_counts, d_expected,
                       n_streams, d_data, data_pitch, data_dw, d_results);
  } else if (aligned4(d_bytes, pitch))
    hipLaunchKernelGGL(scan_kernel<true>, g, b, 0, st, d_bytes, pitch, d_counts, d_expected, n_streams, d_data, data_pitch,
                       d_results);
  else
    hipLaunchKernelGGL(scan_kernel<false>, g, b, 0, st, d_bytes, pitch, d_counts, d_expected, n_streams, d_data,
                       data_pitch, d_results);
  HIP_TRY(hipGetLastError());
  return FSKHIP_OK;
}

int fskhip_xmodem_scan_host(int device, const uint8_t *bytes, size_t pitch, const uint32_t *counts,
                            const uint32_t *expected, uint32_t n_streams, uint8_t *data, size_t data_pitch,
                            fskhip_xmodem_result *results) {
  if (n_streams == 0) return FSKHIP_OK;
  if (!counts || !expected || !results) return fail(FSKHIP_E_INVALID, "fskhip_xmodem_scan_host: null buffer");
  for (uint32_t s = 0; s < n_streams; s++)
    if (counts[s] > pitch) return fail(FSKHIP_E_INVALID, "counts[%u] = %u exceeds pitch %zu", s, counts[s], pitch);
  int rc = select_device(device);
  if (rc != FSKHIP_OK) return rc;
  DevBufs B;
  uint8_t *d_b = nullptr, *d_d = nullptr; uint32_t *d_c = nullptr, *d_e = nullptr; fskhip_xmodem_result *d_r = nullptr;
  const size_t dp = (pitch + 15) & ~(size_t)15;  // 16-byte rows: the tiled kernel applies
  if ((rc = B.alloc(d_b, dp * n_streams)) || (rc = B.alloc(d_d, data_pitch * n_streams)) || (rc = B.alloc(d_c, n_streams)) ||
      (rc = B.alloc(d_e, n_streams)) || (rc = B.alloc(d_r, n_streams)))
    return rc;
  if (pitch) HIP_TRY(hipMemcpy2D(d_b, dp, bytes, pitch, pitch, n_streams, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_c, counts, sizeof(uint32_t) * n_streams, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_e, expected, sizeof(uint32_t) * n_streams, hipMemcpyHostToDevice));
  if (data_pitch) HIP_TRY(hipMemset(d_d, 0, data_pitch * n_streams));
  if ((rc = fskhip_xmodem_scan_device(d_b, dp, d_c, d_e, n_streams, data && data_pitch ? d_d : nullptr, data_pitch, d_r,
                                      nullptr)) != FSKHIP_OK)
    return rc;
  if (data && data_pitch) HIP_TRY(hipMemcpy(data, d_d, data_pitch * n_streams, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(results, d_r, sizeof(fskhip_xmodem_result) * n_streams, hipMemcpyDeviceToHost));
  for (uint32_t s = 0; s < n_streams; s++)
    if (data && results[s].data_len > data_pitch)
      return fail(FSKHIP_E_OVERFLOW, "stream %u assembled %u bytes, slab holds %zu", s, results[s].data_len, data_pitch);
  return FSKHIP_OK;
}

}  // extern "C"
