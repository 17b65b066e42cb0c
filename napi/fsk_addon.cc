// fsk_addon.cc -- thin N-API binding of the C ABI in include/fskhip.h.
//
// This is the FFI a reference maintainer would add: FSKCore / FSKProcessor keep their
// TypeScript signatures and call these functions; nothing here computes DSP.  Built directly against
// /usr/include/node/node_api.h with g++ (no node-gyp download), linked to libfskhip.so.
//
// JS surface (all synchronous; errors throw with the C library's message):
//   create(configs: object | object[], nStreams, device, precision) -> handle
//   destroy(handle)
//   demodulate(handle, samples: Float32Array, nPerStream, pitch, flags) -> {out: Uint8Array, outPitch, counts: Uint32Array, eod: Uint32Array}
//   demodulateAsync(... same ...) -> Promise of the same object; runs on a libuv worker thread
//   modulate(handle, payloads: Uint8Array, lens: Uint32Array, payloadPitch) -> {out: Float32Array, outPitch, lens: Uint32Array}
//   modulatedLength(handle, nBytes) -> number
//   reset(handle, stream)            stream < 0: all
//   getStatus(handle, stream) -> {ready, frameStarted, globalSampleCounter, ...}
//   demodSupported(handle) -> boolean
//   deviceCount() -> number
#include <node_api.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../include/fskhip.h"

#define NAPI_OK(call)                                                        \
  do {                                                                       \
    if ((call) != napi_ok) {                                                 \
      napi_throw_error(env, nullptr, "N-API call failed: " #call);           \
      return nullptr;                                                        \
    }                                                                        \
  } while (0)

static napi_value throw_fsk(napi_env env, int rc) {
  char code[16];
  snprintf(code, sizeof(code), "%d", rc);
  napi_throw_error(env, code, fskhip_last_error());
  return nullptr;
}

static bool get_number(napi_env env, napi_value obj, const char *key, double *out) {
  bool has = false;
  if (napi_has_named_property(env, obj, key, &has) != napi_ok || !has) return false;
  napi_value v;
  if (napi_get_named_property(env, obj, key, &v) != napi_ok) return false;
  napi_valuetype t;
  napi_typeof(env, v, &t);
  if (t == napi_boolean) { bool b; napi_get_value_bool(env, v, &b); *out = b ? 1 : 0; return true; }
  if (t != napi_number) return false;
  return napi_get_value_double(env, v, out) == napi_ok;
}

static bool get_bytes(napi_env env, napi_value obj, const char *key, int32_t *dst, int32_t *len) {
  bool has = false;
  if (napi_has_named_property(env, obj, key, &has) != napi_ok || !has) return true;
  napi_value arr;
  napi_get_named_property(env, obj, key, &arr);
  bool is_arr = false;
  napi_is_array(env, arr, &is_arr);
  if (!is_arr) return false;
  uint32_t n = 0;
  napi_get_array_length(env, arr, &n);
  if (n > FSKHIP_MAX_PATTERN_BYTES) return false;
  for (uint32_t i = 0; i < n; i++) {
    napi_value e;
    napi_get_element(env, arr, i, &e);
    double d = 0;
    napi_get_value_double(env, e, &d);
    dst[i] = (int32_t)d;
  }
  *len = (int32_t)n;
  return true;
}

// FSKConfig object (reference field names, partial objects merge over DEFAULT_FSK_CONFIG like
// configure() does, fsk.ts:134) -> fskhip_config
static bool to_config(napi_env env, napi_value obj, fskhip_config *c) {
  fskhip_default_config(c);
  double d;
  if (get_number(env, obj, "sampleRate", &d)) c->sampleRate = d;
  if (get_number(env, obj, "baudRate", &d)) c->baudRate = d;
  if (get_number(env, obj, "markFrequency", &d)) c->markFrequency = d;
  if (get_number(env, obj, "spaceFrequency", &d)) c->spaceFrequency = d;
  if (get_number(env, obj, "startBits", &d)) c->startBits = (int32_t)d;
  if (get_number(env, obj, "stopBits", &d)) c->stopBits = (int32_t)d;
  if (get_number(env, obj, "syncThreshold", &d)) c->syncThreshold = d;
  if (get_number(env, obj, "agcEnabled", &d)) c->agcEnabled = d != 0;
  if (get_number(env, obj, "preFilterBandwidth", &d)) c->preFilterBandwidth = d;
  if (get_number(env, obj, "adaptiveThreshold", &d)) c->adaptiveThreshold = d != 0;
  if (!get_bytes(env, obj, "preamblePattern", c->preamblePattern, &c->preambleLen)) return false;
  if (!get_bytes(env, obj, "sfdPattern", c->sfdPattern, &c->sfdLen)) return false;
  bool has = false;
  napi_has_named_property(env, obj, "parity", &has);
  if (has) {
    napi_value v;
    napi_get_named_property(env, obj, "parity", &v);
    char buf[16] = {0};
    size_t n = 0;
    if (napi_get_value_string_utf8(env, v, buf, sizeof(buf), &n) == napi_ok) {
      c->parity = !strcmp(buf, "even") ? 1 : !strcmp(buf, "odd") ? 2 : 0;
    }
  }
  return true;
}

// Engines with a demodulateAsync call in flight on a libuv worker (touched on the JS thread only).  EVERY entry point
// that takes the engine handle goes through get_engine(), which refuses a busy engine: the worker is inside
// fskhip_demodulate_host with the engine's staging buffers and host-side counters.  destroy() of a busy engine is
// deferred until its call completes.
static std::vector<fskhip_engine *> g_busy;
static std::vector<fskhip_engine *> g_doomed;
static bool is_busy(const fskhip_engine *e) {
  for (const fskhip_engine *b : g_busy)
    if (b == e) return true;
  return false;
}
static fskhip_engine *get_engine(napi_env env, napi_value v) {
  void *p = nullptr;
  if (napi_get_value_external(env, v, &p) != napi_ok || !p) {
    napi_throw_error(env, nullptr, "FSK modulator not configured");
    return nullptr;
  }
  if (is_busy((fskhip_engine *)p)) {
    napi_throw_error(env, nullptr, "an asynchronous call is already in flight on this engine");
    return nullptr;
  }
  return (fskhip_engine *)p;
}

static napi_value Create(napi_env env, napi_callback_info info) {
  size_t argc = 4;
  napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  if (argc < 4) { napi_throw_type_error(env, nullptr, "create(configs, nStreams, device, precision)"); return nullptr; }
  uint32_t n_streams = 0;
  int32_t device = 0, precision = 0;
  napi_get_value_uint32(env, argv[1], &n_streams);
  napi_get_value_int32(env, argv[2], &device);
  napi_get_value_int32(env, argv[3], &precision);
  std::vector<fskhip_config> cfgs;
  bool is_arr = false;
  napi_is_array(env, argv[0], &is_arr);
  if (is_arr) {
    uint32_t n = 0;
    napi_get_array_length(env, argv[0], &n);
    cfgs.resize(n);
    for (uint32_t i = 0; i < n; i++) {
      napi_value e;
      napi_get_element(env, argv[0], i, &e);
      if (!to_config(env, e, &cfgs[i])) { napi_throw_type_error(env, nullptr, "bad FSKConfig"); return nullptr; }
    }
  } else {
    cfgs.resize(1);
    if (!to_config(env, argv[0], &cfgs[0])) { napi_throw_type_error(env, nullptr, "bad FSKConfig"); return nullptr; }
  }
  fskhip_engine *e = nullptr;
  int rc = fskhip_create(cfgs.data(), (uint32_t)cfgs.size(), n_streams, device, precision, &e);
  if (rc != FSKHIP_OK) return throw_fsk(env, rc);
  napi_value ext;
  NAPI_OK(napi_create_external(env, e, nullptr, nullptr, &ext));
  return ext;
}

static napi_value Destroy(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  void *p = nullptr;
  if (argc == 1 && napi_get_value_external(env, argv[0], &p) == napi_ok && p) {
    if (is_busy((fskhip_engine *)p)) g_doomed.push_back((fskhip_engine *)p);   // destroyed by demod_complete
    else fskhip_destroy((fskhip_engine *)p);
  }
  return nullptr;
}

static napi_value make_typed(napi_env env, napi_typedarray_type t, size_t count, size_t elem, void **data) {
  napi_value ab, ta;
  if (napi_create_arraybuffer(env, count * elem, data, &ab) != napi_ok) return nullptr;
  if (napi_create_typedarray(env, t, count, ab, 0, &ta) != napi_ok) return nullptr;
  return ta;
}

// demodulateData (fsk.ts:190-222) for every stream.  The input Float32Array is BORROWED for the call
// (napi_get_typedarray_info, never retained); with flags & 1 it is overwritten with the AGC-scaled
// samples like the reference does (fsk.ts:55).
static napi_value Demodulate(napi_env env, napi_callback_info info) {
  size_t argc = 5;
  napi_value argv[5];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  fskhip_engine *e = get_engine(env, argv[0]);
  if (!e) return nullptr;
  napi_typedarray_type tt;
  size_t len = 0;
  void *data = nullptr;
  NAPI_OK(napi_get_typedarray_info(env, argv[1], &tt, &len, &data, nullptr, nullptr));
  if (tt != napi_float32_array) { napi_throw_type_error(env, nullptr, "samples must be a Float32Array"); return nullptr; }
  uint32_t n = 0, pitch = 0, flags = 0;
  napi_get_value_uint32(env, argv[2], &n);
  napi_get_value_uint32(env, argv[3], &pitch);
  napi_get_value_uint32(env, argv[4], &flags);
  const uint32_t S = fskhip_n_streams(e);
  if (pitch < n || (size_t)pitch * (S ? S - 1 : 0) + n > len) { napi_throw_range_error(env, nullptr, "samples too short"); return nullptr; }
  const size_t out_pitch = fskhip_max_bytes(e, n);  // the library's own bound (a byte needs >= 8 bit times of samplesPerBit samples)
  void *out = nullptr, *counts = nullptr, *eod = nullptr;
  napi_value out_v = make_typed(env, napi_uint8_array, out_pitch * S, 1, &out);
  napi_value cnt_v = make_typed(env, napi_uint32_array, S, 4, &counts);
  napi_value eod_v = make_typed(env, napi_uint32_array, S, 4, &eod);
  if (!out_v || !cnt_v || !eod_v) { napi_throw_error(env, nullptr, "allocation failed"); return nullptr; }
  int rc = fskhip_demodulate_host(e, (float *)data, n, pitch, (uint8_t *)out, out_pitch, (uint32_t *)counts,
                                  (uint32_t *)eod, flags);
  if (rc != FSKHIP_OK) return throw_fsk(env, rc);
  napi_value res, op;
  NAPI_OK(napi_create_object(env, &res));
  napi_create_uint32(env, (uint32_t)out_pitch, &op);
  napi_set_named_property(env, res, "out", out_v);
  napi_set_named_property(env, res, "outPitch", op);
  napi_set_named_property(env, res, "counts", cnt_v);
  napi_set_named_property(env, res, "eod", eod_v);
  return res;
}

// demodulateAsync: the same call on a libuv worker thread, for batches big enough to matter to the event loop
// (SURVEY 8b "Threading").  Output arrays are created up front on the JS thread; the input Float32Array is
// referenced until the work completes.  One call in flight per engine: a second one rejects.
struct DemodWork {
  napi_async_work work = nullptr;
  napi_deferred deferred = nullptr;
  napi_ref in_ref = nullptr, out_ref = nullptr, cnt_ref = nullptr, eod_ref = nullptr;
  fskhip_engine *e = nullptr;
  float *samples = nullptr;
  uint8_t *out = nullptr;
  uint32_t *counts = nullptr, *eod = nullptr;
  uint32_t n = 0, pitch = 0, flags = 0;
  size_t out_pitch = 0;
  int rc = 0;
  std::string err;
};
static void demod_execute(napi_env, void *data) {
  DemodWork *w = (DemodWork *)data;
  w->rc = fskhip_demodulate_host(w->e, w->samples, w->n, w->pitch, w->out, w->out_pitch, w->counts, w->eod, w->flags);
  if (w->rc != FSKHIP_OK) w->err = fskhip_last_error();  // thread-local: read it on the thread that failed
}
static void demod_complete(napi_env env, napi_status, void *data) {
  DemodWork *w = (DemodWork *)data;
  for (size_t i = 0; i < g_busy.size(); i++)
    if (g_busy[i] == w->e) { g_busy.erase(g_busy.begin() + i); break; }
  bool doomed = false;
  for (size_t i = 0; i < g_doomed.size(); i++)
    if (g_doomed[i] == w->e) { g_doomed.erase(g_doomed.begin() + i); doomed = true; break; }
  if (w->rc == FSKHIP_OK) {
    napi_value res, out_v, cnt_v, eod_v, op;
    napi_create_object(env, &res);
    napi_get_reference_value(env, w->out_ref, &out_v);
    napi_get_reference_value(env, w->cnt_ref, &cnt_v);
    napi_get_reference_value(env, w->eod_ref, &eod_v);
    napi_create_uint32(env, (uint32_t)w->out_pitch, &op);
    napi_set_named_property(env, res, "out", out_v);
    napi_set_named_property(env, res, "outPitch", op);
    napi_set_named_property(env, res, "counts", cnt_v);
    napi_set_named_property(env, res, "eod", eod_v);
    napi_resolve_deferred(env, w->deferred, res);
  } else {
    napi_value msg, errv;
    napi_create_string_utf8(env, w->err.c_str(), NAPI_AUTO_LENGTH, &msg);
    napi_create_error(env, nullptr, msg, &errv);
    napi_reject_deferred(env, w->deferred, errv);
  }
  napi_delete_reference(env, w->in_ref);
  napi_delete_reference(env, w->out_ref);
  napi_delete_reference(env, w->cnt_ref);
  napi_delete_reference(env, w->eod_ref);
  napi_delete_async_work(env, w->work);
  if (doomed) fskhip_destroy(w->e);   // destroy() was called while the worker held the engine
  delete w;
}

static napi_value DemodulateAsync(napi_env env, napi_callback_info info) {
  size_t argc = 5;
  napi_value argv[5];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  fskhip_engine *e = get_engine(env, argv[0]);
  if (!e) return nullptr;
  napi_typedarray_type tt;
  size_t len = 0;
  void *data = nullptr;
  NAPI_OK(napi_get_typedarray_info(env, argv[1], &tt, &len, &data, nullptr, nullptr));
  if (tt != napi_float32_array) { napi_throw_type_error(env, nullptr, "samples must be a Float32Array"); return nullptr; }
  DemodWork *w = new DemodWork();
  w->e = e; w->samples = (float *)data;
  napi_get_value_uint32(env, argv[2], &w->n);
  napi_get_value_uint32(env, argv[3], &w->pitch);
  napi_get_value_uint32(env, argv[4], &w->flags);
  const uint32_t S = fskhip_n_streams(e);
  if (w->pitch < w->n || (size_t)w->pitch * (S ? S - 1 : 0) + w->n > len) { delete w; napi_throw_range_error(env, nullptr, "samples too short"); return nullptr; }
  w->out_pitch = fskhip_max_bytes(e, w->n);
  void *out = nullptr, *counts = nullptr, *eod = nullptr;
  napi_value out_v = make_typed(env, napi_uint8_array, w->out_pitch * S, 1, &out);
  napi_value cnt_v = make_typed(env, napi_uint32_array, S, 4, &counts);
  napi_value eod_v = make_typed(env, napi_uint32_array, S, 4, &eod);
  if (!out_v || !cnt_v || !eod_v) { delete w; napi_throw_error(env, nullptr, "allocation failed"); return nullptr; }
  w->out = (uint8_t *)out; w->counts = (uint32_t *)counts; w->eod = (uint32_t *)eod;
  napi_create_reference(env, argv[1], 1, &w->in_ref);
  napi_create_reference(env, out_v, 1, &w->out_ref);
  napi_create_reference(env, cnt_v, 1, &w->cnt_ref);
  napi_create_reference(env, eod_v, 1, &w->eod_ref);
  napi_value promise, name;
  napi_create_promise(env, &w->deferred, &promise);
  napi_create_string_utf8(env, "fskhip_demodulate", NAPI_AUTO_LENGTH, &name);
  napi_create_async_work(env, nullptr, name, demod_execute, demod_complete, w, &w->work);
  g_busy.push_back(e);
  napi_queue_async_work(env, w->work);
  return promise;
}

// modulateData (fsk.ts:377-424) for every stream
static napi_value Modulate(napi_env env, napi_callback_info info) {
  size_t argc = 4;
  napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  fskhip_engine *e = get_engine(env, argv[0]);
  if (!e) return nullptr;
  napi_typedarray_type tt;
  size_t plen = 0, llen = 0;
  void *pdata = nullptr, *ldata = nullptr;
  NAPI_OK(napi_get_typedarray_info(env, argv[1], &tt, &plen, &pdata, nullptr, nullptr));
  if (tt != napi_uint8_array) { napi_throw_type_error(env, nullptr, "payloads must be a Uint8Array"); return nullptr; }
  NAPI_OK(napi_get_typedarray_info(env, argv[2], &tt, &llen, &ldata, nullptr, nullptr));
  if (tt != napi_uint32_array) { napi_throw_type_error(env, nullptr, "lens must be a Uint32Array"); return nullptr; }
  uint32_t ppitch = 0;
  napi_get_value_uint32(env, argv[3], &ppitch);
  const uint32_t S = fskhip_n_streams(e);
  if (llen < S || plen < (size_t)ppitch * S) { napi_throw_range_error(env, nullptr, "payloads/lens too short"); return nullptr; }
  uint32_t max_len = 0;
  for (uint32_t s = 0; s < S; s++) max_len = ((uint32_t *)ldata)[s] > max_len ? ((uint32_t *)ldata)[s] : max_len;
  size_t out_pitch = fskhip_modulated_length(e, max_len);
  if (out_pitch < 4) out_pitch = 4;
  void *out = nullptr, *olens = nullptr;
  napi_value out_v = make_typed(env, napi_float32_array, out_pitch * S, 4, &out);
  napi_value len_v = make_typed(env, napi_uint32_array, S, 4, &olens);
  if (!out_v || !len_v) { napi_throw_error(env, nullptr, "allocation failed"); return nullptr; }
  int rc = fskhip_modulate_host(e, (const uint8_t *)pdata, (const uint32_t *)ldata, ppitch, (float *)out, out_pitch,
                                (uint32_t *)olens);
  if (rc != FSKHIP_OK) return throw_fsk(env, rc);
  napi_value res, op;
  NAPI_OK(napi_create_object(env, &res));
  napi_create_uint32(env, (uint32_t)out_pitch, &op);
  napi_set_named_property(env, res, "out", out_v);
  napi_set_named_property(env, res, "outPitch", op);
  napi_set_named_property(env, res, "lens", len_v);
  return res;
}

static napi_value ModulatedLength(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  fskhip_engine *e = get_engine(env, argv[0]);
  if (!e) return nullptr;
  uint32_t n = 0;
  napi_get_value_uint32(env, argv[1], &n);
  napi_value r;
  napi_create_double(env, (double)fskhip_modulated_length(e, n), &r);
  return r;
}

static napi_value Reset(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  fskhip_engine *e = get_engine(env, argv[0]);
  if (!e) return nullptr;
  int64_t s = -1;
  napi_get_value_int64(env, argv[1], &s);
  int rc = fskhip_reset(e, s);
  if (rc != FSKHIP_OK) return throw_fsk(env, rc);
  return nullptr;
}

// carryOver(dst, src): what FSKCore.configure() leaves in place on a configured instance (fskhip_carry_over)
static napi_value CarryOver(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  fskhip_engine *dst = get_engine(env, argv[0]);
  if (!dst) return nullptr;
  fskhip_engine *src = get_engine(env, argv[1]);
  if (!src) return nullptr;
  int rc = fskhip_carry_over(dst, src);
  if (rc != FSKHIP_OK) return throw_fsk(env, rc);
  return nullptr;
}

static void set_num(napi_env env, napi_value obj, const char *k, double v);

// enableSignalQuality(handle, on) / getSignalQualityEstimates(handle, stream): the opt-in estimates of include/fskhip.h
static napi_value EnableSignalQuality(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  fskhip_engine *e = get_engine(env, argv[0]);
  if (!e) return nullptr;
  bool on = true;
  if (argc > 1) napi_get_value_bool(env, argv[1], &on);
  int rc = fskhip_enable_signal_quality(e, on ? 1 : 0);
  if (rc != FSKHIP_OK) return throw_fsk(env, rc);
  return nullptr;
}
static napi_value GetSignalQualityEstimates(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  fskhip_engine *e = get_engine(env, argv[0]);
  if (!e) return nullptr;
  uint32_t stream = 0;
  if (argc > 1) napi_get_value_uint32(env, argv[1], &stream);
  fskhip_signal_quality q;
  int rc = fskhip_get_signal_quality(e, stream, &q);
  if (rc != FSKHIP_OK) return throw_fsk(env, rc);
  napi_value obj;
  NAPI_OK(napi_create_object(env, &obj));
  set_num(env, obj, "snr", q.snr); set_num(env, obj, "ber", q.ber); set_num(env, obj, "eyeOpening", q.eyeOpening);
  set_num(env, obj, "phaseJitter", q.phaseJitter); set_num(env, obj, "frequencyOffset", q.frequencyOffset);
  set_num(env, obj, "signalLevel", q.signalLevel); set_num(env, obj, "noiseFloor", q.noiseFloor);
  set_num(env, obj, "frames", q.frames); set_num(env, obj, "bytes", q.bytes);
  return obj;
}

static void set_num(napi_env env, napi_value obj, const char *k, double v) {
  napi_value n;
  napi_create_double(env, v, &n);
  napi_set_named_property(env, obj, k, n);
}
static void set_bool(napi_env env, napi_value obj, const char *k, bool v) {
  napi_value n;
  napi_get_boolean(env, v, &n);
  napi_set_named_property(env, obj, k, n);
}

// getStatus() (fsk.ts:481-493)
static napi_value GetStatus(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  fskhip_engine *e = get_engine(env, argv[0]);
  if (!e) return nullptr;
  uint32_t s = 0;
  napi_get_value_uint32(env, argv[1], &s);
  fskhip_status st;
  int rc = fskhip_get_status(e, s, &st);
  if (rc != FSKHIP_OK) return throw_fsk(env, rc);
  napi_value o;
  NAPI_OK(napi_create_object(env, &o));
  set_bool(env, o, "ready", st.ready != 0);
  set_bool(env, o, "frameStarted", st.frameStarted != 0);
  set_num(env, o, "globalSampleCounter", st.globalSampleCounter);
  set_num(env, o, "receivedBitsLength", st.receivedBitsLength);
  set_num(env, o, "byteBufferLength", st.byteBufferLength);
  set_num(env, o, "demodulationCalls", st.demodulationCalls);
  set_num(env, o, "syncDetections", st.syncDetections);
  set_num(env, o, "silenceThreshold", st.silenceThreshold);
  set_num(env, o, "totalSamplesProcessed", st.totalSamplesProcessed);
  set_num(env, o, "agcGain", st.agcGain);
  set_num(env, o, "eodCount", st.eodCount);
  return o;
}

// getFaults(handle) -> Uint8Array[nStreams]: 1 = the stream's filter state is no longer finite (include/fskhip.h, fskhip_get_faults)
static napi_value GetFaults(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  fskhip_engine *e = get_engine(env, argv[0]);
  if (!e) return nullptr;
  uint32_t n = 0;
  napi_get_value_uint32(env, argv[1], &n);
  void *data = nullptr;
  napi_value ab, arr;
  NAPI_OK(napi_create_arraybuffer(env, n, &data, &ab));
  int rc = fskhip_get_faults(e, n ? static_cast<uint8_t *>(data) : nullptr, nullptr);
  if (rc != FSKHIP_OK) return throw_fsk(env, rc);
  NAPI_OK(napi_create_typedarray(env, napi_uint8_array, n, ab, 0, &arr));
  return arr;
}

static napi_value DemodSupported(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  fskhip_engine *e = get_engine(env, argv[0]);
  if (!e) return nullptr;
  napi_value r;
  napi_get_boolean(env, fskhip_demod_supported(e) != 0, &r);
  return r;
}

static napi_value DeviceCount(napi_env env, napi_callback_info) {
  napi_value r;
  napi_create_int32(env, fskhip_device_count(), &r);
  return r;
}

napi_value InitNext(napi_env env, napi_value exports);  // fsk_addon_next.cc: include/fskhip_next.h

static napi_value Init(napi_env env, napi_value exports) {
  const napi_property_descriptor props[] = {
      {"create", nullptr, Create, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"destroy", nullptr, Destroy, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"demodulate", nullptr, Demodulate, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"demodulateAsync", nullptr, DemodulateAsync, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"modulate", nullptr, Modulate, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"modulatedLength", nullptr, ModulatedLength, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"reset", nullptr, Reset, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"carryOver", nullptr, CarryOver, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"enableSignalQuality", nullptr, EnableSignalQuality, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"getSignalQualityEstimates", nullptr, GetSignalQualityEstimates, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"getStatus", nullptr, GetStatus, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"getFaults", nullptr, GetFaults, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"demodSupported", nullptr, DemodSupported, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"deviceCount", nullptr, DeviceCount, nullptr, nullptr, nullptr, napi_default, nullptr},
  };
  napi_define_properties(env, exports, sizeof(props) / sizeof(props[0]), props);
  napi_value v;
  napi_create_int32(env, fskhip_abi_version(), &v);
  napi_set_named_property(env, exports, "abiVersion", v);
  return InitNext(env, exports);
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
