// fsk_addon_next.cc -- N-API binding of include/fskhip_next.h (CRC-16 / XModem packets, the FSKProcessor streaming
// contract, batched FIR), part of fsk_addon.node.  Same rules as fsk_addon.cc: typed arrays are borrowed for the
// call, results are fresh arrays, errors throw with the C library's message, nothing here computes.
//
// JS surface added to the addon:
//   crc16(data: Uint8Array, pitch, lens: Uint32Array, device) -> Uint16Array
//   xmodemSerialize(payloads: Uint8Array, pitch, lens: Uint32Array, seqs: Uint32Array, device) -> {out, outPitch, lens}
//   xmodemScan(bytes: Uint8Array, pitch, counts: Uint32Array, expected: Uint32Array, device) -> {data, dataPitch, results: Int32Array[10*S]}
//   processorCreate(engine, rxCapacity) -> handle;  processorDestroy(handle)
//   processorProcess(handle, input: Float32Array|null, nIn, inPitch, nOut, flags) -> Float32Array|null
//   processorModulate(handle, payloads: Uint8Array, lens: Uint32Array, pitch, mask: Uint8Array|null)
//   processorTxState(handle) -> {pos, total, pending, completed};  processorRxLength(handle) -> Uint32Array
//   processorDrain(handle, capacity) -> {out, outPitch, counts};  processorReset(handle, stream)
//   sincLowpass/sincHighpass(cutoff, sampleRate, numTaps), sincBandpass(center, bandwidth, sampleRate, numTaps) -> Float64Array
//   firCreate(taps: Float64Array, nStreams, device, precision) -> handle;  firDestroy(handle)
//   firProcess(handle, input: Float32Array, n, pitch, nStreams) -> Float32Array;  firReset(handle, stream)
//   iirCreate(b, a: Float64Array, nStreams, device, precision) -> handle; iirProcess(handle, Float32Array | Float64Array, n, pitch, nStreams);
//   iirCoefficients(handle), iirReset(handle, stream), iirDestroy(handle); butterworth(kind, f, bandwidth, sampleRate) -> [b0 b1 b2 a0 a1 a2]
#include <node_api.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../include/fskhip_next.h"

#define NAPI_OK(call)                                                        \
  do {                                                                       \
    if ((call) != napi_ok) {                                                 \
      napi_throw_error(env, nullptr, "N-API call failed: " #call);           \
      return nullptr;                                                        \
    }                                                                        \
  } while (0)

namespace {

napi_value throw_fsk(napi_env env, int rc) {
  char code[16];
  snprintf(code, sizeof(code), "%d", rc);
  napi_throw_error(env, code, fskhip_last_error());
  return nullptr;
}
napi_value make_typed(napi_env env, napi_typedarray_type t, size_t count, size_t elem, void **data) {
  napi_value ab, ta;
  if (napi_create_arraybuffer(env, count * elem, data, &ab) != napi_ok) return nullptr;
  if (napi_create_typedarray(env, t, count, ab, 0, &ta) != napi_ok) return nullptr;
  return ta;
}
// borrows a typed array of the given type; null/undefined -> data = nullptr when `optional`
bool typed(napi_env env, napi_value v, napi_typedarray_type want, void **data, size_t *len, bool optional = false) {
  napi_valuetype vt;
  napi_typeof(env, v, &vt);
  if (optional && (vt == napi_null || vt == napi_undefined)) { *data = nullptr; *len = 0; return true; }
  bool is = false;
  napi_is_typedarray(env, v, &is);
  napi_typedarray_type tt;
  if (!is || napi_get_typedarray_info(env, v, &tt, len, data, nullptr, nullptr) != napi_ok || tt != want) {
    napi_throw_type_error(env, nullptr, "wrong typed array argument");
    return false;
  }
  return true;
}
uint32_t u32(napi_env env, napi_value v) { uint32_t x = 0; napi_get_value_uint32(env, v, &x); return x; }
int32_t i32(napi_env env, napi_value v) { int32_t x = 0; napi_get_value_int32(env, v, &x); return x; }
double f64(napi_env env, napi_value v) { double x = 0; napi_get_value_double(env, v, &x); return x; }
void *external(napi_env env, napi_value v, const char *what) {
  void *p = nullptr;
  if (napi_get_value_external(env, v, &p) != napi_ok || !p) { napi_throw_error(env, nullptr, what); return nullptr; }
  return p;
}
// FIR / IIR filter handles: the external owns a box, so that close() can empty it (a later call throws instead of touching freed
// memory) and a filter that is never close()d is destroyed when the garbage collector drops the handle
struct FilterBox { void *p; bool iir; };
void filter_box_finalize(napi_env, void *data, void *) {
  FilterBox *b = (FilterBox *)data;
  if (b->p) { if (b->iir) fskhip_iir_destroy((fskhip_iir *)b->p); else fskhip_fir_destroy((fskhip_fir *)b->p); }
  delete b;
}
napi_value filter_box_new(napi_env env, void *p, bool iir) {
  FilterBox *b = new FilterBox{p, iir};
  napi_value ext;
  if (napi_create_external(env, b, filter_box_finalize, nullptr, &ext) != napi_ok) { filter_box_finalize(env, b, nullptr); return nullptr; }
  return ext;
}
void *filter_of(napi_env env, napi_value v) {
  FilterBox *b = (FilterBox *)external(env, v, "filter destroyed");
  if (!b) return nullptr;
  if (!b->p) { napi_throw_error(env, nullptr, "filter destroyed"); return nullptr; }
  return b->p;
}
void filter_box_close(napi_env env, napi_value v) {
  void *q = nullptr;
  if (napi_get_value_external(env, v, &q) != napi_ok || !q) return;
  FilterBox *b = (FilterBox *)q;
  if (b->p) { if (b->iir) fskhip_iir_destroy((fskhip_iir *)b->p); else fskhip_fir_destroy((fskhip_fir *)b->p); b->p = nullptr; }
}
void set_u32(napi_env env, napi_value obj, const char *k, uint32_t v) {
  napi_value n;
  napi_create_uint32(env, v, &n);
  napi_set_named_property(env, obj, k, n);
}

#define ARGS(n)                                                              \
  size_t argc = n;                                                           \
  napi_value argv[n];                                                        \
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));       \
  if (argc < n) { napi_throw_type_error(env, nullptr, "too few arguments"); return nullptr; }

napi_value Crc16(napi_env env, napi_callback_info info) {
  ARGS(4);
  void *data, *lens; size_t dlen, llen;
  if (!typed(env, argv[0], napi_uint8_array, &data, &dlen) || !typed(env, argv[2], napi_uint32_array, &lens, &llen)) return nullptr;
  const uint32_t pitch = u32(env, argv[1]);
  if ((size_t)pitch * llen > dlen) { napi_throw_range_error(env, nullptr, "data too short"); return nullptr; }
  void *out;
  napi_value out_v = make_typed(env, napi_uint16_array, llen, 2, &out);
  int rc = fskhip_crc16_host(i32(env, argv[3]), (const uint8_t *)data, pitch, (const uint32_t *)lens, (uint32_t)llen, (uint16_t *)out);
  if (rc != FSKHIP_OK) return throw_fsk(env, rc);
  return out_v;
}

napi_value XmodemSerialize(napi_env env, napi_callback_info info) {
  ARGS(5);
  void *pl, *lens, *seqs; size_t plen, llen, slen;
  if (!typed(env, argv[0], napi_uint8_array, &pl, &plen) || !typed(env, argv[2], napi_uint32_array, &lens, &llen) ||
      !typed(env, argv[3], napi_uint32_array, &seqs, &slen)) return nullptr;
  const uint32_t pitch = u32(env, argv[1]);
  if (slen < llen || (size_t)pitch * llen > plen) { napi_throw_range_error(env, nullptr, "arguments too short"); return nullptr; }
  uint32_t mx = 0;
  for (size_t i = 0; i < llen; i++) mx = ((uint32_t *)lens)[i] > mx ? ((uint32_t *)lens)[i] : mx;
  const size_t out_pitch = (size_t)mx + 6;
  void *out, *olens;
  napi_value out_v = make_typed(env, napi_uint8_array, out_pitch * llen, 1, &out);
  napi_value len_v = make_typed(env, napi_uint32_array, llen, 4, &olens);
  int rc = fskhip_xmodem_serialize_host(i32(env, argv[4]), (const uint8_t *)pl, pitch, (const uint32_t *)lens, (const uint32_t *)seqs,
                                        (uint32_t)llen, (uint8_t *)out, out_pitch, (uint32_t *)olens);
  if (rc != FSKHIP_OK) return throw_fsk(env, rc);
  napi_value res;
  NAPI_OK(napi_create_object(env, &res));
  napi_set_named_property(env, res, "out", out_v);
  napi_set_named_property(env, res, "lens", len_v);
  set_u32(env, res, "outPitch", (uint32_t)out_pitch);
  return res;
}

napi_value XmodemScan(napi_env env, napi_callback_info info) {
  ARGS(5);
  void *bytes, *counts, *expected; size_t blen, clen, elen;
  if (!typed(env, argv[0], napi_uint8_array, &bytes, &blen) || !typed(env, argv[2], napi_uint32_array, &counts, &clen) ||
      !typed(env, argv[3], napi_uint32_array, &expected, &elen)) return nullptr;
  const uint32_t pitch = u32(env, argv[1]);
  if (elen < clen || (size_t)pitch * clen > blen) { napi_throw_range_error(env, nullptr, "arguments too short"); return nullptr; }
  const size_t data_pitch = pitch ? pitch : 4;
  static_assert(sizeof(fskhip_xmodem_result) == 40, "result layout");
  void *data, *res;
  napi_value data_v = make_typed(env, napi_uint8_array, data_pitch * clen, 1, &data);
  napi_value res_v = make_typed(env, napi_int32_array, 10 * clen, 4, &res);
  int rc = fskhip_xmodem_scan_host(i32(env, argv[4]), (const uint8_t *)bytes, pitch, (const uint32_t *)counts, (const uint32_t *)expected,
                                   (uint32_t)clen, (uint8_t *)data, data_pitch, (fskhip_xmodem_result *)res);
  if (rc != FSKHIP_OK) return throw_fsk(env, rc);
  napi_value o;
  NAPI_OK(napi_create_object(env, &o));
  napi_set_named_property(env, o, "data", data_v);
  napi_set_named_property(env, o, "results", res_v);
  set_u32(env, o, "dataPitch", (uint32_t)data_pitch);
  return o;
}

// ---- FSKProcessor -----------------------------------------------------------------------------------
struct Proc { fskhip_processor *p; uint32_t S; };

napi_value ProcessorCreate(napi_env env, napi_callback_info info) {
  ARGS(2);
  fskhip_engine *e = (fskhip_engine *)external(env, argv[0], "FSK modulator not configured");
  if (!e) return nullptr;
  fskhip_processor *p = nullptr;
  int rc = fskhip_processor_create(e, u32(env, argv[1]), &p);
  if (rc != FSKHIP_OK) return throw_fsk(env, rc);
  Proc *h = new Proc{p, fskhip_n_streams(e)};
  napi_value ext;
  NAPI_OK(napi_create_external(env, h, nullptr, nullptr, &ext));
  return ext;
}
napi_value ProcessorDestroy(napi_env env, napi_callback_info info) {
  ARGS(1);
  void *p = nullptr;
  if (napi_get_value_external(env, argv[0], &p) == napi_ok && p) {
    Proc *h = (Proc *)p;
    if (h->p) fskhip_processor_destroy(h->p);
    h->p = nullptr;
  }
  return nullptr;
}
Proc *get_proc(napi_env env, napi_value v) {
  Proc *h = (Proc *)external(env, v, "processor destroyed");
  if (h && !h->p) { napi_throw_error(env, nullptr, "processor destroyed"); return nullptr; }
  return h;
}

napi_value ProcessorProcess(napi_env env, napi_callback_info info) {
  ARGS(6);
  Proc *h = get_proc(env, argv[0]);
  if (!h) return nullptr;
  void *in; size_t ilen;
  if (!typed(env, argv[1], napi_float32_array, &in, &ilen, true)) return nullptr;
  const uint32_t n_in = u32(env, argv[2]), in_pitch = u32(env, argv[3]), n_out = u32(env, argv[4]), flags = u32(env, argv[5]);
  if (in && (in_pitch < n_in || (size_t)in_pitch * (h->S - 1) + n_in > ilen)) { napi_throw_range_error(env, nullptr, "input too short"); return nullptr; }
  void *out = nullptr;
  napi_value out_v = nullptr;
  if (n_out) out_v = make_typed(env, napi_float32_array, (size_t)n_out * h->S, 4, &out);
  int rc = fskhip_processor_process_host(h->p, (float *)in, n_in, in_pitch, (float *)out, n_out, n_out, flags);
  if (rc != FSKHIP_OK) return throw_fsk(env, rc);
  if (!out_v) napi_get_null(env, &out_v);
  return out_v;
}

napi_value ProcessorModulate(napi_env env, napi_callback_info info) {
  ARGS(5);
  Proc *h = get_proc(env, argv[0]);
  if (!h) return nullptr;
  void *pl, *lens, *mask; size_t plen, llen, mlen;
  if (!typed(env, argv[1], napi_uint8_array, &pl, &plen) || !typed(env, argv[2], napi_uint32_array, &lens, &llen) ||
      !typed(env, argv[4], napi_uint8_array, &mask, &mlen, true)) return nullptr;
  const uint32_t pitch = u32(env, argv[3]);
  if (llen < h->S || (size_t)pitch * h->S > plen || (mask && mlen < h->S)) { napi_throw_range_error(env, nullptr, "arguments too short"); return nullptr; }
  int rc = fskhip_processor_modulate_host(h->p, (const uint8_t *)pl, (const uint32_t *)lens, pitch, (const uint8_t *)mask);
  if (rc != FSKHIP_OK) return throw_fsk(env, rc);
  return nullptr;
}

napi_value ProcessorTxState(napi_env env, napi_callback_info info) {
  ARGS(1);
  Proc *h = get_proc(env, argv[0]);
  if (!h) return nullptr;
  void *pos, *total, *pending, *completed;
  napi_value pos_v = make_typed(env, napi_uint32_array, h->S, 4, &pos), tot_v = make_typed(env, napi_uint32_array, h->S, 4, &total);
  napi_value pen_v = make_typed(env, napi_uint8_array, h->S, 1, &pending), com_v = make_typed(env, napi_uint32_array, h->S, 4, &completed);
  int rc = fskhip_processor_tx_state_host(h->p, (uint32_t *)pos, (uint32_t *)total, (uint8_t *)pending, (uint32_t *)completed);
  if (rc != FSKHIP_OK) return throw_fsk(env, rc);
  napi_value o;
  NAPI_OK(napi_create_object(env, &o));
  napi_set_named_property(env, o, "pos", pos_v);
  napi_set_named_property(env, o, "total", tot_v);
  napi_set_named_property(env, o, "pending", pen_v);
  napi_set_named_property(env, o, "completed", com_v);
  return o;
}

napi_value ProcessorRxLength(napi_env env, napi_callback_info info) {
  ARGS(1);
  Proc *h = get_proc(env, argv[0]);
  if (!h) return nullptr;
  void *lens;
  napi_value v = make_typed(env, napi_uint32_array, h->S, 4, &lens);
  int rc = fskhip_processor_rx_length_host(h->p, (uint32_t *)lens);
  if (rc != FSKHIP_OK) return throw_fsk(env, rc);
  return v;
}

napi_value ProcessorDrain(napi_env env, napi_callback_info info) {
  ARGS(2);
  Proc *h = get_proc(env, argv[0]);
  if (!h) return nullptr;
  const uint32_t cap = u32(env, argv[1]);
  void *out, *counts;
  napi_value out_v = make_typed(env, napi_uint8_array, (size_t)cap * h->S, 1, &out);
  napi_value cnt_v = make_typed(env, napi_uint32_array, h->S, 4, &counts);
  int rc = fskhip_processor_rx_drain_host(h->p, (uint8_t *)out, cap, (uint32_t *)counts);
  if (rc != FSKHIP_OK) return throw_fsk(env, rc);
  napi_value o;
  NAPI_OK(napi_create_object(env, &o));
  napi_set_named_property(env, o, "out", out_v);
  napi_set_named_property(env, o, "counts", cnt_v);
  set_u32(env, o, "outPitch", cap);
  return o;
}

napi_value ProcessorReset(napi_env env, napi_callback_info info) {
  ARGS(2);
  Proc *h = get_proc(env, argv[0]);
  if (!h) return nullptr;
  int64_t s = -1;
  napi_get_value_int64(env, argv[1], &s);
  int rc = fskhip_processor_reset(h->p, s);
  if (rc != FSKHIP_OK) return throw_fsk(env, rc);
  return nullptr;
}

// ---- FIR ------------------------------------------------------------------------------------------------
napi_value sinc_result(napi_env env, int n, const std::vector<double> &t) {
  if (n < 0) return throw_fsk(env, n);
  void *out;
  napi_value v = make_typed(env, napi_float64_array, (size_t)n, 8, &out);
  memcpy(out, t.data(), sizeof(double) * (size_t)n);
  return v;
}
napi_value SincLowpass(napi_env env, napi_callback_info info) {
  ARGS(3);
  const uint32_t nt = u32(env, argv[2]);
  std::vector<double> t(nt + 2);
  return sinc_result(env, fskhip_sinc_lowpass(f64(env, argv[0]), f64(env, argv[1]), nt, t.data()), t);
}
napi_value SincHighpass(napi_env env, napi_callback_info info) {
  ARGS(3);
  const uint32_t nt = u32(env, argv[2]);
  std::vector<double> t(nt + 2);
  return sinc_result(env, fskhip_sinc_highpass(f64(env, argv[0]), f64(env, argv[1]), nt, t.data()), t);
}
napi_value SincBandpass(napi_env env, napi_callback_info info) {
  ARGS(4);
  const uint32_t nt = u32(env, argv[3]);
  std::vector<double> t(nt + 2);
  return sinc_result(env, fskhip_sinc_bandpass(f64(env, argv[0]), f64(env, argv[1]), f64(env, argv[2]), nt, t.data()), t);
}

napi_value FirCreate(napi_env env, napi_callback_info info) {
  ARGS(4);
  void *taps; size_t nt;
  if (!typed(env, argv[0], napi_float64_array, &taps, &nt)) return nullptr;
  fskhip_fir *f = nullptr;
  int rc = fskhip_fir_create(i32(env, argv[2]), (const double *)taps, (uint32_t)nt, u32(env, argv[1]), i32(env, argv[3]), &f);
  if (rc != FSKHIP_OK) return throw_fsk(env, rc);
  return filter_box_new(env, f, false);
}
napi_value FirDestroy(napi_env env, napi_callback_info info) {
  ARGS(1);
  filter_box_close(env, argv[0]);
  return nullptr;
}
napi_value FirProcess(napi_env env, napi_callback_info info) {
  ARGS(5);
  fskhip_fir *f = (fskhip_fir *)filter_of(env, argv[0]);
  if (!f) return nullptr;
  void *in; size_t ilen;
  if (!typed(env, argv[1], napi_float32_array, &in, &ilen)) return nullptr;
  const uint32_t n = u32(env, argv[2]), pitch = u32(env, argv[3]), S = u32(env, argv[4]);
  if (S != fskhip_fir_streams(f)) { napi_throw_range_error(env, nullptr, "nStreams is not the filter's stream count"); return nullptr; }
  if (S == 0 || pitch < n || (size_t)pitch * (S - 1) + n > ilen) { napi_throw_range_error(env, nullptr, "input too short"); return nullptr; }
  void *out;
  napi_value out_v = make_typed(env, napi_float32_array, (size_t)n * S, 4, &out);
  int rc = fskhip_fir_process_host(f, (const float *)in, n, pitch, (float *)out, n);
  if (rc != FSKHIP_OK) return throw_fsk(env, rc);
  return out_v;
}
napi_value FirReset(napi_env env, napi_callback_info info) {
  ARGS(2);
  fskhip_fir *f = (fskhip_fir *)filter_of(env, argv[0]);
  if (!f) return nullptr;
  int64_t s = -1;
  napi_get_value_int64(env, argv[1], &s);
  int rc = fskhip_fir_reset(f, s);
  if (rc != FSKHIP_OK) return throw_fsk(env, rc);
  return nullptr;
}

// ---- IIRFilter (src/dsp/filters.ts:8-106) ----
napi_value IirCreate(napi_env env, napi_callback_info info) {   // (b: Float64Array, a: Float64Array, nStreams, device, precision)
  ARGS(5);
  void *b, *a; size_t nb, na;
  if (!typed(env, argv[0], napi_float64_array, &b, &nb) || !typed(env, argv[1], napi_float64_array, &a, &na)) return nullptr;
  fskhip_iir *f = nullptr;
  int rc = fskhip_iir_create(i32(env, argv[3]), (const double *)b, (uint32_t)nb, (const double *)a, (uint32_t)na, u32(env, argv[2]), i32(env, argv[4]), &f);
  if (rc != FSKHIP_OK) return throw_fsk(env, rc);
  return filter_box_new(env, f, true);
}
napi_value IirDestroy(napi_env env, napi_callback_info info) {
  ARGS(1);
  filter_box_close(env, argv[0]);
  return nullptr;
}
napi_value IirCoefficients(napi_env env, napi_callback_info info) {   // -> Float64Array [nb, na, b..., a...]
  ARGS(1);
  fskhip_iir *f = (fskhip_iir *)filter_of(env, argv[0]);
  if (!f) return nullptr;
  double b[9], a[9];
  uint32_t nb = 0, na = 0;
  int rc = fskhip_iir_get_coefficients(f, b, &nb, a, &na);
  if (rc != FSKHIP_OK) return throw_fsk(env, rc);
  void *out;
  napi_value out_v = make_typed(env, napi_float64_array, 2 + nb + na, 8, &out);
  double *o = (double *)out;
  o[0] = nb; o[1] = na;
  for (uint32_t i = 0; i < nb; i++) o[2 + i] = b[i];
  for (uint32_t i = 0; i < na; i++) o[2 + nb + i] = a[i];
  return out_v;
}
napi_value IirProcess(napi_env env, napi_callback_info info) {   // Float32Array in -> Float32Array (processBuffer); Float64Array in -> Float64Array (process)
  ARGS(5);
  fskhip_iir *f = (fskhip_iir *)filter_of(env, argv[0]);
  if (!f) return nullptr;
  napi_typedarray_type ty; size_t ilen; void *in; napi_value ab; size_t off;
  if (napi_get_typedarray_info(env, argv[1], &ty, &ilen, &in, &ab, &off) != napi_ok || (ty != napi_float32_array && ty != napi_float64_array)) {
    napi_throw_type_error(env, nullptr, "Float32Array or Float64Array expected");
    return nullptr;
  }
  const uint32_t n = u32(env, argv[2]), pitch = u32(env, argv[3]), S = u32(env, argv[4]);
  if (S != fskhip_iir_streams(f)) { napi_throw_range_error(env, nullptr, "nStreams is not the filter's stream count"); return nullptr; }
  if (S == 0 || pitch < n || (size_t)pitch * (S - 1) + n > ilen) { napi_throw_range_error(env, nullptr, "input too short"); return nullptr; }
  void *out;
  int rc;
  napi_value out_v;
  if (ty == napi_float32_array) {
    out_v = make_typed(env, napi_float32_array, (size_t)n * S, 4, &out);
    rc = fskhip_iir_process_host(f, (const float *)in, n, pitch, (float *)out, n);
  } else {
    out_v = make_typed(env, napi_float64_array, (size_t)n * S, 8, &out);
    rc = fskhip_iir_process_f64_host(f, (const double *)in, n, pitch, (double *)out, n);
  }
  if (rc != FSKHIP_OK) return throw_fsk(env, rc);
  return out_v;
}
napi_value IirReset(napi_env env, napi_callback_info info) {
  ARGS(2);
  fskhip_iir *f = (fskhip_iir *)filter_of(env, argv[0]);
  if (!f) return nullptr;
  int64_t s = -1;
  napi_get_value_int64(env, argv[1], &s);
  int rc = fskhip_iir_reset(f, s);
  if (rc != FSKHIP_OK) return throw_fsk(env, rc);
  return nullptr;
}
napi_value Butterworth(napi_env env, napi_callback_info info) {   // (kind 0 lowpass | 1 highpass | 2 bandpass, f, [bandwidth,] sampleRate) -> Float64Array [b0 b1 b2 a0 a1 a2]
  ARGS(4);
  const int kind = i32(env, argv[0]);
  double b[3], a[3];
  if (kind == 0) fskhip_butterworth_lowpass(f64(env, argv[1]), f64(env, argv[3]), b, a);
  else if (kind == 1) fskhip_butterworth_highpass(f64(env, argv[1]), f64(env, argv[3]), b, a);
  else fskhip_butterworth_bandpass(f64(env, argv[1]), f64(env, argv[2]), f64(env, argv[3]), b, a);
  void *out;
  napi_value out_v = make_typed(env, napi_float64_array, 6, 8, &out);
  double *o = (double *)out;
  for (int i = 0; i < 3; i++) { o[i] = b[i]; o[3 + i] = a[i]; }
  return out_v;
}

}  // namespace

napi_value InitNext(napi_env env, napi_value exports) {
  const napi_property_descriptor props[] = {
      {"crc16", nullptr, Crc16, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"xmodemSerialize", nullptr, XmodemSerialize, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"xmodemScan", nullptr, XmodemScan, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"processorCreate", nullptr, ProcessorCreate, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"processorDestroy", nullptr, ProcessorDestroy, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"processorProcess", nullptr, ProcessorProcess, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"processorModulate", nullptr, ProcessorModulate, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"processorTxState", nullptr, ProcessorTxState, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"processorRxLength", nullptr, ProcessorRxLength, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"processorDrain", nullptr, ProcessorDrain, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"processorReset", nullptr, ProcessorReset, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"sincLowpass", nullptr, SincLowpass, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"sincHighpass", nullptr, SincHighpass, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"sincBandpass", nullptr, SincBandpass, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"firCreate", nullptr, FirCreate, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"firDestroy", nullptr, FirDestroy, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"firProcess", nullptr, FirProcess, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"firReset", nullptr, FirReset, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"iirCreate", nullptr, IirCreate, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"iirDestroy", nullptr, IirDestroy, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"iirCoefficients", nullptr, IirCoefficients, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"iirProcess", nullptr, IirProcess, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"iirReset", nullptr, IirReset, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"butterworth", nullptr, Butterworth, nullptr, nullptr, nullptr, napi_default, nullptr},
  };
  napi_define_properties(env, exports, sizeof(props) / sizeof(props[0]), props);
  return exports;
}
