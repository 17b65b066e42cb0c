// fsk_processor.hip -- C ABI of the FSKProcessor / ChunkedModulator streaming contract (include/fskhip_next.h,
// SURVEY.md 8(f1)): per-stream RX byte ring and pending modulation resident on the device, one process() per
// quantum = the demodulator launch(es) + one bookkeeping/TX launch, optionally replayed as a captured hipGraph
// (a 128-sample quantum is launch-bound, not bandwidth-bound).
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "fsk_host.h"
#include "fsk_params.h"

namespace fsk {
hipError_t launch_processor_io(const ModParams &M, const double *coef, const ProcState &T, const uint8_t *demod_out,
                               size_t demod_pitch, const uint32_t *demod_counts, bool do_rx, float *out, size_t n_out,
                               size_t out_pitch, bool clear_rx_on_complete, hipStream_t st);
hipError_t launch_processor_tx_start(const ModParams &M, const ProcState &T, const uint8_t *payloads, const uint32_t *lens,
                                     size_t payload_pitch, const uint8_t *mask, hipStream_t st);
hipError_t launch_processor_rx_drain(const ProcState &T, uint32_t n_streams, uint8_t *out, size_t out_pitch,
                                     uint32_t *counts, hipStream_t st);
hipError_t launch_processor_reset(const ProcState &T, uint32_t n_streams, int64_t stream, bool rx, bool tx, hipStream_t st);
// engine internals the processor needs (fsk_api.hip)
const ModParams &engine_mod_params(const fskhip_engine *e);
const double *engine_coef(const fskhip_engine *e);
size_t engine_max_bytes(const fskhip_engine *e, size_t n_per_stream);
uint32_t engine_launch_key(const fskhip_engine *e);       // changes whenever the demodulator would launch differently
void engine_refresh_kernel_choice(fskhip_engine *e);     // "blk_resets" = auto: look at the tile statistics the last completed call left
void engine_note_replayed_call(fskhip_engine *e, size_t n);  // host-side counters of a call replayed from a graph
}  // namespace fsk

using namespace fsk;

struct fskhip_processor {
  fskhip_engine *e = nullptr;
  int device = 0;
  uint32_t S = 0;
  ProcState T{};
  // demodulator outputs of the current quantum
  uint8_t *d_bytes = nullptr; size_t bytes_pitch = 0;
  uint32_t *d_counts = nullptr, *d_eod = nullptr;
  // staging for the _host entry points
  hipStream_t stream = nullptr;
  float *d_in = nullptr; size_t d_in_cap = 0;
  float *d_out = nullptr; size_t d_out_cap = 0;
  uint8_t *d_stage = nullptr; size_t d_stage_cap = 0;
  uint32_t *d_u32 = nullptr;   // [4][S] scratch
  uint8_t *d_mask = nullptr;
  // captured quantum
  hipGraphExec_t graph_exec = nullptr;
  struct Key {
    float *in; size_t n_in, in_pitch; float *out; size_t n_out, out_pitch; uint32_t flags; hipStream_t st; uint32_t ekey;
    bool operator==(const Key &o) const {
      return in == o.in && n_in == o.n_in && in_pitch == o.in_pitch && out == o.out && n_out == o.n_out &&
             out_pitch == o.out_pitch && flags == o.flags && st == o.st && ekey == o.ekey;
    }
  } graph_key{};
};

namespace {

template <typename T>
int dev_alloc(T *&p, size_t n) {
  hipError_t err = hipMalloc((void **)&p, (n ? n : 1) * sizeof(T));
  if (err != hipSuccess) return fail(FSKHIP_E_NOMEM, "hipMalloc(%zu): %s", n * sizeof(T), hipGetErrorString(err));
  return FSKHIP_OK;
}
template <typename T>
int ensure(T *&p, size_t &cap, size_t need) {
  if (need <= cap) return FSKHIP_OK;
  if (p) (void)hipFree(p);
  p = nullptr; cap = 0;
  int rc = dev_alloc(p, need);
  if (rc == FSKHIP_OK) cap = need;
  return rc;
}

void drop_graph(fskhip_processor *p) {
  if (p->graph_exec) (void)hipGraphExecDestroy(p->graph_exec);
  p->graph_exec = nullptr;
}

// the launches of one quantum, in stream order
int launch_quantum(fskhip_processor *p, float *d_in, size_t n_in, size_t in_pitch, float *d_out, size_t n_out,
                   size_t out_pitch, uint32_t flags, hipStream_t st) {
  if (d_in) {
    int rc = fskhip_demodulate_device(p->e, d_in, n_in, in_pitch, p->d_bytes, p->bytes_pitch, p->d_counts, p->d_eod, 0u, st);
    if (rc != FSKHIP_OK) return rc;
  }
  HIP_TRY(launch_processor_io(engine_mod_params(p->e), engine_coef(p->e), p->T, p->d_bytes, p->bytes_pitch, p->d_counts,
                              d_in != nullptr, d_out, n_out, out_pitch, (flags & FSKHIP_PROC_CLEAR_RX_ON_TX_COMPLETE) != 0, st));
  return FSKHIP_OK;
}

}  // namespace

extern "C" {

int fskhip_processor_destroy(fskhip_processor *p) {
  if (!p) return FSKHIP_OK;
  (void)hipSetDevice(p->device);
  (void)hipDeviceSynchronize();
  drop_graph(p);
  void *bufs[] = {p->T.rx_buf, p->T.rx_w, p->T.rx_r, p->T.rx_len, p->T.tx_payload, p->T.tx_phase, p->T.tx_pos, p->T.tx_len,
                  p->T.tx_in_bit, p->T.tx_bit_idx, p->T.tx_cur_bit, p->T.tx_n_payload, p->T.tx_pending, p->T.tx_completed,
                  p->d_bytes, p->d_counts, p->d_eod, p->d_in, p->d_out, p->d_stage, p->d_u32, p->d_mask};
  for (void *b : bufs)
    if (b) (void)hipFree(b);
  if (p->stream) (void)hipStreamDestroy(p->stream);
  delete p;
  return FSKHIP_OK;
}

int fskhip_processor_create(fskhip_engine *e, uint32_t rx_capacity, fskhip_processor **out) {
  if (!e || !out) return fail(FSKHIP_E_INVALID, "fskhip_processor_create: null argument");
  if (rx_capacity == 0) return fail(FSKHIP_E_INVALID, "rx_capacity must be > 0");
  fskhip_processor *p = new (std::nothrow) fskhip_processor();
  if (!p) return fail(FSKHIP_E_NOMEM, "out of host memory");
  p->e = e; p->device = engine_device(e); p->S = fskhip_n_streams(e);
  const size_t S = p->S;
  ProcState &T = p->T;
  T.rx_cap = rx_capacity;
  int rc = FSKHIP_OK;
  hipError_t herr = hipSetDevice(p->device);
  if (herr != hipSuccess) { delete p; return fail(FSKHIP_E_HIP, "hipSetDevice: %s", hipGetErrorString(herr)); }
#define PROC_TRY(expr)                                   \
  do {                                                   \
    if (rc == FSKHIP_OK) rc = (expr);                    \
  } while (0)
  PROC_TRY(dev_alloc(T.rx_buf, S * rx_capacity));
  PROC_TRY(dev_alloc(T.rx_w, S)); PROC_TRY(dev_alloc(T.rx_r, S)); PROC_TRY(dev_alloc(T.rx_len, S));
  PROC_TRY(dev_alloc(T.tx_phase, S));
  PROC_TRY(dev_alloc(T.tx_pos, S)); PROC_TRY(dev_alloc(T.tx_len, S)); PROC_TRY(dev_alloc(T.tx_in_bit, S));
  PROC_TRY(dev_alloc(T.tx_bit_idx, S)); PROC_TRY(dev_alloc(T.tx_cur_bit, S)); PROC_TRY(dev_alloc(T.tx_n_payload, S));
  PROC_TRY(dev_alloc(T.tx_pending, S)); PROC_TRY(dev_alloc(T.tx_completed, S));
  PROC_TRY(dev_alloc(p->d_counts, S)); PROC_TRY(dev_alloc(p->d_eod, S)); PROC_TRY(dev_alloc(p->d_u32, 4 * S));
  PROC_TRY(dev_alloc(p->d_mask, S));
#undef PROC_TRY
  if (rc == FSKHIP_OK) {
    uint32_t *zero[] = {T.rx_w, T.rx_r, T.rx_len, T.tx_pos, T.tx_len, T.tx_in_bit, T.tx_bit_idx, T.tx_cur_bit,
                        T.tx_n_payload, T.tx_pending, T.tx_completed, p->d_counts, p->d_eod};
    for (uint32_t *z : zero)
      if (hipMemset(z, 0, sizeof(uint32_t) * S) != hipSuccess) rc = fail(FSKHIP_E_HIP, "hipMemset failed");
    if (hipMemset(T.tx_phase, 0, sizeof(double) * S) != hipSuccess) rc = fail(FSKHIP_E_HIP, "hipMemset failed");
    if (hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) != hipSuccess) rc = fail(FSKHIP_E_HIP, "hipStreamCreate failed");
    if (hipDeviceSynchronize() != hipSuccess) rc = fail(FSKHIP_E_HIP, "hipDeviceSynchronize failed");
  }
  if (rc != FSKHIP_OK) {
    std::string keep = fskhip_last_error();
    fskhip_processor_destroy(p);
    return fail(rc, "%s", keep.c_str());
  }
  *out = p;
  return FSKHIP_OK;
}

int fskhip_processor_process_device(fskhip_processor *p, float *d_in, size_t n_in, size_t in_pitch, float *d_out,
                                    size_t n_out, size_t out_pitch, uint32_t flags, void *hip_stream) {
  if (!p) return fail(FSKHIP_E_INVALID, "null processor");
  if (d_in && in_pitch < n_in) return fail(FSKHIP_E_INVALID, "in_pitch %zu < n_in %zu", in_pitch, n_in);
  if (d_out && out_pitch < n_out) return fail(FSKHIP_E_INVALID, "out_pitch %zu < n_out %zu", out_pitch, n_out);
  if (d_in && !fskhip_demod_supported(p->e)) {
    // let the engine produce its own loud message
    return fskhip_demodulate_device(p->e, d_in, n_in, in_pitch, p->d_bytes, p->bytes_pitch, p->d_counts, p->d_eod, 0u, hip_stream);
  }
  HIP_TRY(hipSetDevice(p->device));
  hipStream_t st = (hipStream_t)hip_stream;
  if (d_in) {  // byte slab of this quantum: grown outside any capture
    const size_t need = engine_max_bytes(p->e, n_in);
    if (need > p->bytes_pitch) {
      HIP_TRY(hipDeviceSynchronize());
      drop_graph(p);
      if (p->d_bytes) (void)hipFree(p->d_bytes);
      p->d_bytes = nullptr; p->bytes_pitch = 0;
      int rc = dev_alloc(p->d_bytes, need * p->S);
      if (rc != FSKHIP_OK) return rc;
      p->bytes_pitch = need;
    }
  }
  // timing events / the trace capture are per-launch host decisions: no replay while either is armed
  engine_refresh_kernel_choice(p->e);
  if (engine_launch_key(p->e) & (8u | 16u)) flags &= ~FSKHIP_PROC_GRAPH;
  if (!(flags & FSKHIP_PROC_GRAPH)) return launch_quantum(p, d_in, n_in, in_pitch, d_out, n_out, out_pitch, flags, st);

  if (!st) return fail(FSKHIP_E_INVALID, "FSKHIP_PROC_GRAPH needs an explicit stream (the null stream cannot be captured)");
  fskhip_processor::Key key{d_in, n_in, in_pitch, d_out, n_out, out_pitch, flags, st, engine_launch_key(p->e)};
  if (!p->graph_exec || !(key == p->graph_key)) {
    drop_graph(p);
    hipGraph_t graph = nullptr;
    HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    int rc = launch_quantum(p, d_in, n_in, in_pitch, d_out, n_out, out_pitch, flags, st);
    hipError_t cerr = hipStreamEndCapture(st, &graph);
    if (rc != FSKHIP_OK) {
      if (graph) (void)hipGraphDestroy(graph);
      return rc;
    }
    if (cerr != hipSuccess) return fail(FSKHIP_E_HIP, "hipStreamEndCapture: %s", hipGetErrorString(cerr));
    hipError_t ierr = hipGraphInstantiate(&p->graph_exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ierr != hipSuccess) { p->graph_exec = nullptr; return fail(FSKHIP_E_HIP, "hipGraphInstantiate: %s", hipGetErrorString(ierr)); }
    p->graph_key = key;
    // the capture itself already did the host-side accounting of one call; it launched nothing
    HIP_TRY(hipGraphLaunch(p->graph_exec, st));
    return FSKHIP_OK;
  }
  if (d_in) engine_note_replayed_call(p->e, n_in);
  HIP_TRY(hipGraphLaunch(p->graph_exec, st));
  return FSKHIP_OK;
}

int fskhip_processor_process_host(fskhip_processor *p, float *in, size_t n_in, size_t in_pitch, float *out, size_t n_out,
                                  size_t out_pitch, uint32_t flags) {
  if (!p) return fail(FSKHIP_E_INVALID, "null processor");
  if (in && in_pitch < n_in) return fail(FSKHIP_E_INVALID, "in_pitch %zu < n_in %zu", in_pitch, n_in);
  if (out && out_pitch < n_out) return fail(FSKHIP_E_INVALID, "out_pitch %zu < n_out %zu", out_pitch, n_out);
  HIP_TRY(hipSetDevice(p->device));
  const size_t S = p->S;
  const size_t ip = (n_in + 3) & ~(size_t)3, op = (n_out + 3) & ~(size_t)3;
  int rc;
  if (in && (rc = ensure(p->d_in, p->d_in_cap, (ip ? ip : 4) * S)) != FSKHIP_OK) return rc;
  if (out && (rc = ensure(p->d_out, p->d_out_cap, (op ? op : 4) * S)) != FSKHIP_OK) return rc;
  if (in && n_in)
    HIP_TRY(hipMemcpy2DAsync(p->d_in, ip * sizeof(float), in, in_pitch * sizeof(float), n_in * sizeof(float), S,
                             hipMemcpyHostToDevice, p->stream));
  rc = fskhip_processor_process_device(p, in ? p->d_in : nullptr, n_in, ip ? ip : 4, out ? p->d_out : nullptr, n_out,
                                       op ? op : 4, flags, p->stream);
  if (rc != FSKHIP_OK) return rc;
  if (out && n_out)
    HIP_TRY(hipMemcpy2DAsync(out, out_pitch * sizeof(float), p->d_out, op * sizeof(float), n_out * sizeof(float), S,
                             hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return FSKHIP_OK;
}

int fskhip_processor_modulate_host(fskhip_processor *p, const uint8_t *payloads, const uint32_t *lens, size_t payload_pitch,
                                   const uint8_t *mask) {
  if (!p) return fail(FSKHIP_E_INVALID, "null processor");
  if (!lens) return fail(FSKHIP_E_INVALID, "null lens");
  HIP_TRY(hipSetDevice(p->device));
  HIP_TRY(hipDeviceSynchronize());
  const size_t S = p->S;
  std::vector<uint32_t> pending(S);
  HIP_TRY(hipMemcpy(pending.data(), p->T.tx_pending, sizeof(uint32_t) * S, hipMemcpyDeviceToHost));
  size_t max_len = 0;
  for (size_t s = 0; s < S; s++) {
    if (mask && !mask[s]) continue;
    if (pending[s]) return fail(FSKHIP_E_BUSY, "Modulation already in progress (stream %zu)", s);
    if (lens[s] > payload_pitch) return fail(FSKHIP_E_INVALID, "lens[%zu] = %u exceeds payload_pitch %zu", s, lens[s], payload_pitch);
    if (lens[s] && !payloads) return fail(FSKHIP_E_INVALID, "null payloads");
    if (lens[s] > max_len) max_len = lens[s];
  }
  if (max_len > p->T.tx_payload_pitch) {  // grow the payload store, keeping the pending rows
    const size_t np = (max_len + 63) & ~(size_t)63;
    uint8_t *nbuf = nullptr;
    int rc = dev_alloc(nbuf, np * S);
    if (rc != FSKHIP_OK) return rc;
    HIP_TRY(hipMemset(nbuf, 0, np * S));
    if (p->T.tx_payload) {
      HIP_TRY(hipMemcpy2D(nbuf, np, p->T.tx_payload, p->T.tx_payload_pitch, p->T.tx_payload_pitch, S, hipMemcpyDeviceToDevice));
      (void)hipFree(p->T.tx_payload);
    }
    p->T.tx_payload = nbuf;
    p->T.tx_payload_pitch = np;
    drop_graph(p);  // the captured launch holds the old pointer
  }
  int rc;
  if ((rc = ensure(p->d_stage, p->d_stage_cap, (payload_pitch ? payload_pitch : 1) * S)) != FSKHIP_OK) return rc;
  if (payload_pitch && payloads) HIP_TRY(hipMemcpy(p->d_stage, payloads, payload_pitch * S, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(p->d_u32, lens, sizeof(uint32_t) * S, hipMemcpyHostToDevice));
  if (mask) HIP_TRY(hipMemcpy(p->d_mask, mask, S, hipMemcpyHostToDevice));
  HIP_TRY(launch_processor_tx_start(engine_mod_params(p->e), p->T, p->d_stage, p->d_u32, payload_pitch,
                                    mask ? p->d_mask : nullptr, nullptr));
  HIP_TRY(hipDeviceSynchronize());
  return FSKHIP_OK;
}

int fskhip_processor_tx_state_host(fskhip_processor *p, uint32_t *pos, uint32_t *total, uint8_t *pending, uint32_t *completed) {
  if (!p) return fail(FSKHIP_E_INVALID, "null processor");
  HIP_TRY(hipSetDevice(p->device));
  HIP_TRY(hipDeviceSynchronize());
  const size_t S = p->S;
  if (pos) HIP_TRY(hipMemcpy(pos, p->T.tx_pos, sizeof(uint32_t) * S, hipMemcpyDeviceToHost));
  if (total) HIP_TRY(hipMemcpy(total, p->T.tx_len, sizeof(uint32_t) * S, hipMemcpyDeviceToHost));
  if (completed) HIP_TRY(hipMemcpy(completed, p->T.tx_completed, sizeof(uint32_t) * S, hipMemcpyDeviceToHost));
  if (pending) {
    std::vector<uint32_t> tmp(S);
    HIP_TRY(hipMemcpy(tmp.data(), p->T.tx_pending, sizeof(uint32_t) * S, hipMemcpyDeviceToHost));
    for (size_t s = 0; s < S; s++) pending[s] = tmp[s] ? 1 : 0;
  }
  return FSKHIP_OK;
}

int fskhip_processor_rx_drain_host(fskhip_processor *p, uint8_t *out, size_t out_pitch, uint32_t *counts) {
  if (!p) return fail(FSKHIP_E_INVALID, "null processor");
  if (!counts || (out_pitch && !out)) return fail(FSKHIP_E_INVALID, "null buffer");
  HIP_TRY(hipSetDevice(p->device));
  HIP_TRY(hipDeviceSynchronize());
  const size_t S = p->S;
  int rc;
  if ((rc = ensure(p->d_stage, p->d_stage_cap, (out_pitch ? out_pitch : 1) * S)) != FSKHIP_OK) return rc;
  HIP_TRY(launch_processor_rx_drain(p->T, p->S, p->d_stage, out_pitch, p->d_u32, nullptr));
  HIP_TRY(hipMemcpy(counts, p->d_u32, sizeof(uint32_t) * S, hipMemcpyDeviceToHost));
  if (out_pitch) HIP_TRY(hipMemcpy(out, p->d_stage, out_pitch * S, hipMemcpyDeviceToHost));
  for (size_t s = 0; s < S; s++)
    if (counts[s] > out_pitch) return fail(FSKHIP_E_OVERFLOW, "stream %zu held %u bytes, slab holds %zu", s, counts[s], out_pitch);
  return FSKHIP_OK;
}

int fskhip_processor_rx_length_host(fskhip_processor *p, uint32_t *lengths) {
  if (!p || !lengths) return fail(FSKHIP_E_INVALID, "null argument");
  HIP_TRY(hipSetDevice(p->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(lengths, p->T.rx_len, sizeof(uint32_t) * p->S, hipMemcpyDeviceToHost));
  return FSKHIP_OK;
}

int fskhip_processor_reset(fskhip_processor *p, int64_t stream) {
  if (!p) return fail(FSKHIP_E_INVALID, "null processor");
  if (stream >= (int64_t)p->S) return fail(FSKHIP_E_INVALID, "stream out of range");
  HIP_TRY(hipSetDevice(p->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(launch_processor_reset(p->T, p->S, stream, true, true, nullptr));
  HIP_TRY(hipDeviceSynchronize());
  return FSKHIP_OK;
}

}  // extern "C"
