/*
 * fsk_oracle.c -- scalar fp64 restatement of the reference FSK hot path.
 * TEST INFRASTRUCTURE: see fsk_oracle.h.  Build with -O2 -ffp-contract=off (no FMA
 * contraction: the reference is plain IEEE double arithmetic in JavaScript).
 *
 * Every function cites the reference lines it restates (src/modems/fsk.ts,
 * src/dsp/filters.ts, src/utils.ts).
 */
#include "fsk_oracle.h"
#include "v8_sin.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#ifndef M_SQRT2
#define M_SQRT2 1.41421356237309504880
#endif

/* ------------------------------------------------------------------------------------------
 * RingBuffer<TypedArray>  (utils.ts:6-105)
 *
 * The reference constructs the sync-bit ring with a possibly FRACTIONAL size
 * (maxSyncBits * dsSPB * 1.1, fsk.ts:149).  JavaScript semantics are kept: indices are
 * doubles, the typed array has floor(size) elements, stores to a non-integer or
 * out-of-range index are dropped and loads from one yield `undefined` (UNDEF here).
 * ---------------------------------------------------------------------------------------- */
#define UNDEF (-1)

typedef struct ring {
  double max_length;   /* utils.ts:11 maxLength (may be fractional) */
  long arr_len;        /* typed array length = ToIndex(size) */
  double read_index, write_index, length;
  uint8_t *u8;         /* exactly one of u8 / f32 is used */
  float *f32;
} ring;

static int ring_init(ring *r, double size, int is_float) {
  memset(r, 0, sizeof(*r));
  r->max_length = size;
  r->arr_len = (long)floor(size);
  if (r->arr_len < 0) r->arr_len = 0;
  if (is_float) r->f32 = (float *)calloc((size_t)r->arr_len + 1, sizeof(float));
  else r->u8 = (uint8_t *)calloc((size_t)r->arr_len + 1, 1);
  return (r->f32 || r->u8) ? 0 : -1;
}
static void ring_free(ring *r) { free(r->u8); free(r->f32); r->u8 = NULL; r->f32 = NULL; }
static int is_index(double x, long n) { return x >= 0 && x == floor(x) && x < (double)n; }

/* put() utils.ts:38-48 */
static void ring_put_u8(ring *r, int v) {
  if (is_index(r->write_index, r->arr_len)) r->u8[(long)r->write_index] = (uint8_t)v;
  r->write_index = fmod(r->write_index + 1, r->max_length);
  if (r->length < r->max_length) r->length += 1;
  else r->read_index = fmod(r->read_index + 1, r->max_length);
}
static void ring_put_f32(ring *r, double v) {
  if (is_index(r->write_index, r->arr_len)) r->f32[(long)r->write_index] = (float)v; /* Float32Array store */
  r->write_index = fmod(r->write_index + 1, r->max_length);
  if (r->length < r->max_length) r->length += 1;
  else r->read_index = fmod(r->read_index + 1, r->max_length);
}
/* get() utils.ts:28-36; the bounds `throw` cannot trigger from FSKCore's call sites */
static int ring_get_u8(const ring *r, double index) {
  double p;
  if (index < 0) index += r->length;
  p = fmod(r->read_index + index, r->max_length);
  return is_index(p, r->arr_len) ? (int)r->u8[(long)p] : UNDEF;
}
static double ring_get_f32(const ring *r, double index) {
  double p;
  if (index < 0) index += r->length;
  p = fmod(r->read_index + index, r->max_length);
  return is_index(p, r->arr_len) ? (double)r->f32[(long)p] : NAN; /* undefined -> NaN in a sum */
}
static void ring_clear(ring *r) { r->read_index = r->write_index = r->length = 0; } /* utils.ts:92-96 */

/* ------------------------------------------------------------------------------------------
 * IIRFilter  (filters.ts:8-106)
 * ---------------------------------------------------------------------------------------- */
struct fsko_iir {
  int nb, na, order, xlen, ylen, x_index, y_index;
  double *b, *a, *x, *y;
};

/* reset() filters.ts:92-98 */
void fsko_iir_reset(fsko_iir *f) {
  memset(f->x, 0, sizeof(double) * (size_t)f->xlen);
  memset(f->y, 0, sizeof(double) * (size_t)(f->ylen > 0 ? f->ylen : 1));
  f->x_index = 0;
  f->y_index = 0;
}

/* constructor filters.ts:17-42 */
fsko_iir *fsko_iir_create(const double *b, int nb, const double *a, int na) {
  fsko_iir *f;
  int i;
  if (!b || nb <= 0) return NULL;     /* 'Feedforward coefficients (b) cannot be empty' */
  if (!a || na <= 0) return NULL;     /* 'Feedback coefficients (a) cannot be empty' */
  if (a[0] == 0) return NULL;         /* 'First feedback coefficient (a[0]) cannot be zero' */
  f = (fsko_iir *)calloc(1, sizeof(*f));
  if (!f) return NULL;
  f->nb = nb; f->na = na;
  f->order = (nb > na ? nb : na) - 1;
  f->xlen = nb > f->order + 1 ? nb : f->order + 1;
  f->ylen = (na - 1) > f->order ? (na - 1) : f->order;
  f->b = (double *)malloc(sizeof(double) * (size_t)nb);
  f->a = (double *)malloc(sizeof(double) * (size_t)na);
  f->x = (double *)malloc(sizeof(double) * (size_t)f->xlen);
  f->y = (double *)malloc(sizeof(double) * (size_t)(f->ylen > 0 ? f->ylen : 1));
  memcpy(f->b, b, sizeof(double) * (size_t)nb);
  memcpy(f->a, a, sizeof(double) * (size_t)na);
  if (f->a[0] != 1) { /* filters.ts:30-39 */
    double a0 = f->a[0];
    for (i = 0; i < nb; i++) f->b[i] /= a0;
    for (i = 1; i < na; i++) f->a[i] /= a0;
    f->a[0] = 1;
  }
  fsko_iir_reset(f);
  return f;
}
void fsko_iir_destroy(fsko_iir *f) {
  if (!f) return;
  free(f->b); free(f->a); free(f->x); free(f->y); free(f);
}
void fsko_iir_coefficients(const fsko_iir *f, double *b, double *a) {
  memcpy(b, f->b, sizeof(double) * (size_t)f->nb);
  memcpy(a, f->a, sizeof(double) * (size_t)f->na);
}

/* process() filters.ts:47-76: Direct Form I, accumulated left to right from 0 */
double fsko_iir_process(fsko_iir *f, double input) {
  double output = 0;
  int i, x_idx, y_idx;
  f->x[f->x_index] = input;
  x_idx = f->x_index;
  for (i = 0; i < f->nb; i++) {
    output += f->b[i] * f->x[x_idx];
    x_idx = x_idx == 0 ? f->xlen - 1 : x_idx - 1;
  }
  if (f->ylen > 0) {
    y_idx = f->y_index == 0 ? f->ylen - 1 : f->y_index - 1;
    for (i = 1; i < f->na; i++) {
      output -= f->a[i] * f->y[y_idx];
      y_idx = y_idx == 0 ? f->ylen - 1 : y_idx - 1;
    }
    f->y[f->y_index] = output;
    f->y_index = (f->y_index + 1) % f->ylen;
  }
  f->x_index = (f->x_index + 1) % f->xlen;
  return output;
}
/* processBuffer() filters.ts:81-87: Float32Array in, new Float32Array out (f32 store) */
void fsko_iir_process_buffer(fsko_iir *f, const float *in, float *out, size_t n) {
  size_t i;
  for (i = 0; i < n; i++) out[i] = (float)fsko_iir_process(f, (double)in[i]);
}

/* ------------------------------------------------------------------------------------------
 * FIRFilter  (filters.ts:112-167)
 * ---------------------------------------------------------------------------------------- */
struct fsko_fir {
  int n, index;
  double *c, *delay;
};
fsko_fir *fsko_fir_create(const double *taps, int n) {
  fsko_fir *f = (fsko_fir *)calloc(1, sizeof(*f));
  if (!f || n <= 0) { free(f); return NULL; }
  f->n = n;
  f->c = (double *)malloc(sizeof(double) * (size_t)n);
  f->delay = (double *)calloc((size_t)n, sizeof(double));
  memcpy(f->c, taps, sizeof(double) * (size_t)n);
  return f;
}
void fsko_fir_destroy(fsko_fir *f) { if (f) { free(f->c); free(f->delay); free(f); } }
void fsko_fir_reset(fsko_fir *f) { memset(f->delay, 0, sizeof(double) * (size_t)f->n); f->index = 0; }
/* process() filters.ts:125-140 */
double fsko_fir_process(fsko_fir *f, double input) {
  double output = 0;
  int i, d = f->index;
  f->delay[f->index] = input;
  for (i = 0; i < f->n; i++) {
    output += f->c[i] * f->delay[d];
    d = d == 0 ? f->n - 1 : d - 1;
  }
  f->index = (f->index + 1) % f->n;
  return output;
}
void fsko_fir_process_buffer(fsko_fir *f, const float *in, float *out, size_t n) {
  size_t i;
  for (i = 0; i < n; i++) out[i] = (float)fsko_fir_process(f, (double)in[i]);
}

/* ------------------------------------------------------------------------------------------
 * FilterDesign  (filters.ts:172-315)
 * ---------------------------------------------------------------------------------------- */
/* butterworthLowpass filters.ts:180-192 */
void fsko_butterworth_lowpass(double cutoff, double sampleRate, double b[3], double a[3]) {
  double nyquist = sampleRate / 2;
  double normalizedCutoff = cutoff / nyquist;
  double c = tan(M_PI * normalizedCutoff / 2);
  double c2 = c * c;
  double sqrt2c = M_SQRT2 * c;
  double denom = 1 + sqrt2c + c2;
  b[0] = c2 / denom; b[1] = 2 * c2 / denom; b[2] = c2 / denom;
  a[0] = 1; a[1] = (2 * c2 - 2) / denom; a[2] = (1 - sqrt2c + c2) / denom;
}
/* butterworthHighpass filters.ts:200-212 */
void fsko_butterworth_highpass(double cutoff, double sampleRate, double b[3], double a[3]) {
  double nyquist = sampleRate / 2;
  double normalizedCutoff = cutoff / nyquist;
  double c = tan(M_PI * normalizedCutoff / 2);
  double c2 = c * c;
  double sqrt2c = M_SQRT2 * c;
  double denom = 1 + sqrt2c + c2;
  b[0] = 1 / denom; b[1] = -2 / denom; b[2] = 1 / denom;
  a[0] = 1; a[1] = (2 * c2 - 2) / denom; a[2] = (1 - sqrt2c + c2) / denom;
}
/* butterworthBandpass filters.ts:221-234 */
void fsko_butterworth_bandpass(double center, double bandwidth, double sampleRate, double b[3], double a[3]) {
  double omega = 2 * M_PI * center / sampleRate;
  double bw = 2 * M_PI * bandwidth / sampleRate;
  double c = tan(bw / 2);
  double d = 2 * cos(omega);
  double c2 = c * c;
  double denom = 1 + c + c2;
  b[0] = c / denom; b[1] = 0; b[2] = -c / denom;
  a[0] = 1; a[1] = (-d * (1 + c2)) / denom; a[2] = (1 - c + c2) / denom;
}
/* sincLowpass filters.ts:243-265 */
int fsko_sinc_lowpass(double cutoff, double sampleRate, int numTaps, double *taps) {
  double normalizedCutoff, center;
  int i;
  if (numTaps % 2 == 0) numTaps++;
  normalizedCutoff = cutoff / sampleRate;
  center = (numTaps - 1) / 2.0;
  for (i = 0; i < numTaps; i++) {
    if ((double)i == center) {
      taps[i] = 2 * normalizedCutoff;
    } else {
      double x = M_PI * (i - center);
      taps[i] = sin(2 * normalizedCutoff * x) / x;
    }
    taps[i] *= 0.54 - 0.46 * cos(2 * M_PI * i / (numTaps - 1));
  }
  return numTaps;
}
/* sincHighpass filters.ts:274-286.  NOTE: the reference passes the caller's (possibly even)
 * numTaps on to the loop bounds while sincLowpass returns an odd-length array; restated as is
 * for odd numTaps, which is all the reference's factories and tests use. */
int fsko_sinc_highpass(double cutoff, double sampleRate, int numTaps, double *taps) {
  int n = fsko_sinc_lowpass(cutoff, sampleRate, numTaps, taps);
  int i, center = (numTaps - 1) / 2;
  for (i = 0; i < numTaps && i < n; i++) taps[i] = -taps[i];
  if (numTaps % 2 == 1) taps[center] += 1; /* even numTaps: lowpass[x.5] += 1 touches no element */
  return n;
}
/* sincBandpass filters.ts:296-314: truncated convolution of a highpass and a lowpass */
int fsko_sinc_bandpass(double center, double bandwidth, double sampleRate, int numTaps, double *taps) {
  double lowFreq = center - bandwidth / 2;
  double highFreq = center + bandwidth / 2;
  double *hp = (double *)malloc(sizeof(double) * (size_t)(numTaps + 1));
  double *lp = (double *)malloc(sizeof(double) * (size_t)(numTaps + 1));
  int i, j;
  fsko_sinc_highpass(lowFreq, sampleRate, numTaps, hp);
  fsko_sinc_lowpass(highFreq, sampleRate, numTaps, lp);
  for (i = 0; i < numTaps; i++) taps[i] = 0;
  for (i = 0; i < numTaps; i++)
    for (j = 0; j < numTaps; j++)
      if (i + j < numTaps) taps[i + j] += hp[i] * lp[j];
  free(hp); free(lp);
  return numTaps;
}

/* ------------------------------------------------------------------------------------------
 * FSKCore  (fsk.ts:82-494) and AGCProcessor (fsk.ts:38-77)
 * ---------------------------------------------------------------------------------------- */
#define MAX_PATTERN_BITS (FSKO_MAX_PATTERN_BYTES * 2 * 32)

struct fsko_core {
  int ready;
  fsko_config cfg;
  /* params fsk.ts:95-99 */
  double samplesPerBit, bitsPerByte, centerFreq, downsampleRatio, downsampledSamplesPerBit;
  /* AGC fsk.ts:39-42 */
  int has_agc;
  double agc_target, agc_gain, agc_attack, agc_release;
  /* dsp fsk.ts:87-92 */
  fsko_iir *pre, *lp_i, *lp_q, *post;
  /* iqState fsk.ts:102, downsample fsk.ts:105-109 */
  double local_osc_phase, last_phase;
  double ds_counter, ds_i, ds_q;
  /* bitSync fsk.ts:112-115 */
  double global_sample_counter, bit_sample_counter, bit_accumulator, bit_accum_count, next_bit_sample_index;
  /* frame fsk.ts:118-122 */
  int pattern[MAX_PATTERN_BITS + 1];
  int n_pattern;
  double max_sync_bits;
  int started;
  ring sync_ring, amp_ring;
  int rings_ok;
  /* byteState fsk.ts:125 */
  int byte_current, bit_position;
  uint8_t *out;
  size_t out_cap, out_n;
  /* silence fsk.ts:128 */
  double silence_threshold, samples_for_eod, silence_count;
  /* debug fsk.ts:131 */
  double sync_detections, demod_calls, total_samples;
  double eod_count;
  fsko_trace *trace;
  /* signal-quality estimates (fsk_oracle.h: NOT the reference's getSignalQuality(), which returns zeros) */
  int q_on, q_armed, q_prev_d0;
  double q_last_post, q_vote_ones, q_vote_count;
  double q_signal, q_floor, q_frames, q_f_sum, q_f2_sum, q_eye_sum, q_bytes, q_minor, q_votes, q_f0_sum, q_starts, q_ftrans;
};

void fsko_default_config(fsko_config *cfg) { /* fsk.ts:19-33 */
  memset(cfg, 0, sizeof(*cfg));
  cfg->sampleRate = 48000;
  cfg->baudRate = 1200;
  cfg->markFrequency = 1650;
  cfg->spaceFrequency = 1850;
  cfg->preamblePattern[0] = 0x55; cfg->preamblePattern[1] = 0x55; cfg->preambleLen = 2;
  cfg->sfdPattern[0] = 0x7E; cfg->sfdLen = 1;
  cfg->startBits = 1;
  cfg->stopBits = 1;
  cfg->parity = 0;
  cfg->syncThreshold = 0.85;
  cfg->agcEnabled = 1;
  cfg->preFilterBandwidth = 800;
  cfg->adaptiveThreshold = 1;
}

fsko_core *fsko_create(void) {
  fsko_core *c = (fsko_core *)calloc(1, sizeof(*c));
  if (c) c->silence_threshold = 0.01; /* fsk.ts:128: set at construction only */
  return c;
}

static void free_dsp(fsko_core *c) {
  fsko_iir_destroy(c->pre); fsko_iir_destroy(c->lp_i); fsko_iir_destroy(c->lp_q); fsko_iir_destroy(c->post);
  c->pre = c->lp_i = c->lp_q = c->post = NULL;
  if (c->rings_ok) { ring_free(&c->sync_ring); ring_free(&c->amp_ring); c->rings_ok = 0; }
}
void fsko_destroy(fsko_core *c) {
  if (!c) return;
  free_dsp(c);
  free(c->out);
  free(c);
}
void fsko_set_trace(fsko_core *c, fsko_trace *tr) { c->trace = tr; }

/* addByteToPattern fsk.ts:159-173 */
static void add_byte_to_pattern(fsko_core *c, int byte) {
  int i;
  for (i = 0; i < c->cfg.startBits; i++) c->pattern[c->n_pattern++] = 0;
  for (i = 7; i >= 0; i--) c->pattern[c->n_pattern++] = (byte >> i) & 1;
  if (c->cfg.parity != 0) {
    int parity = 0;
    for (i = 0; i < 8; i++) parity ^= (byte >> i) & 1;
    c->pattern[c->n_pattern++] = c->cfg.parity == 1 ? parity : 1 - parity;
  }
  for (i = 0; i < c->cfg.stopBits; i++) c->pattern[c->n_pattern++] = 1;
}

/* resetState fsk.ts:175-188 */
static void reset_state(fsko_core *c) {
  c->local_osc_phase = 0; c->last_phase = 0;
  c->global_sample_counter = 0; c->bit_sample_counter = 0; c->bit_accumulator = 0;
  c->bit_accum_count = 0; c->next_bit_sample_index = 0;
  c->byte_current = 0; c->bit_position = 0;
  c->started = 0;
  c->silence_count = 0;
  if (c->lp_i) fsko_iir_reset(c->lp_i);
  if (c->lp_q) fsko_iir_reset(c->lp_q);
  if (c->post) fsko_iir_reset(c->post);
  c->ds_counter = 0; c->ds_i = 0; c->ds_q = 0;
}

int fsko_configure(fsko_core *c, const fsko_config *cfg) { /* fsk.ts:133-157 */
  double b[3], a[3], downsampleRate, freqSpan, deviation, carson, finalBw;
  int i;
  if (cfg->preambleLen < 0 || cfg->preambleLen > FSKO_MAX_PATTERN_BYTES) return -1;
  if (cfg->sfdLen < 0 || cfg->sfdLen > FSKO_MAX_PATTERN_BYTES) return -1;
  if (cfg->startBits < 0 || cfg->startBits > 8 || cfg->stopBits < 0 || cfg->stopBits > 8) return -1;
  free_dsp(c);
  c->cfg = *cfg;
  /* calculateParameters fsk.ts:426-444 */
  c->downsampleRatio = 2;
  downsampleRate = cfg->sampleRate / c->downsampleRatio;
  c->centerFreq = (cfg->markFrequency + cfg->spaceFrequency) / 2;
  c->samplesPerBit = floor(cfg->sampleRate / cfg->baudRate);
  c->bitsPerByte = 8 + cfg->startBits + cfg->stopBits + (cfg->parity != 0 ? 1 : 0);
  c->downsampledSamplesPerBit = floor(downsampleRate / cfg->baudRate);
  /* initializeDSP fsk.ts:446-462 */
  c->has_agc = cfg->agcEnabled ? 1 : 0;
  if (c->has_agc) { /* AGCProcessor ctor fsk.ts:44-50 */
    c->agc_target = 0.5;
    c->agc_gain = 1.0;
    c->agc_attack = 1.0 - exp(-1.0 / (cfg->sampleRate * 0.001));
    c->agc_release = 1.0 - exp(-1.0 / (cfg->sampleRate * 0.01));
  }
  freqSpan = fabs(cfg->spaceFrequency - cfg->markFrequency);
  deviation = freqSpan / 2;
  carson = 2 * (deviation + cfg->baudRate);
  finalBw = cfg->preFilterBandwidth > carson ? cfg->preFilterBandwidth : carson; /* Math.max */
  fsko_butterworth_bandpass(c->centerFreq, finalBw, cfg->sampleRate, b, a);
  c->pre = fsko_iir_create(b, 3, a, 3);
  fsko_butterworth_lowpass(cfg->baudRate, cfg->sampleRate, b, a);
  c->lp_i = fsko_iir_create(b, 3, a, 3);
  c->lp_q = fsko_iir_create(b, 3, a, 3);
  c->post = fsko_iir_create(b, 3, a, 3); /* designed for sampleRate, run at sampleRate/2 (fsk.ts:461) */
  if (!c->pre || !c->lp_i || !c->lp_q || !c->post) return -1;
  /* frame detection fsk.ts:143-145 */
  c->n_pattern = 0;
  for (i = 0; i < cfg->preambleLen; i++) add_byte_to_pattern(c, cfg->preamblePattern[i]);
  for (i = 0; i < cfg->sfdLen; i++) add_byte_to_pattern(c, cfg->sfdPattern[i]);
  c->pattern[c->n_pattern] = UNDEF; /* preambleSfdBits[length] is `undefined` (fsk.ts:307, j = 0) */
  c->max_sync_bits = c->n_pattern + 32;
  /* fsk.ts:148-150 */
  c->samples_for_eod = c->bitsPerByte * c->downsampledSamplesPerBit * 0.7;
  if (ring_init(&c->sync_ring, c->max_sync_bits * c->downsampledSamplesPerBit * 1.1, 0)) return -1;
  if (ring_init(&c->amp_ring, c->downsampledSamplesPerBit * 8, 1)) return -1;
  c->rings_ok = 1;
  reset_state(c);
  c->ready = 1;
  return 0;
}

static void push_byte(fsko_core *c, int v) {
  if (c->out_n == c->out_cap) {
    size_t ncap = c->out_cap ? c->out_cap * 2 : 256;
    c->out = (uint8_t *)realloc(c->out, ncap);
    c->out_cap = ncap;
  }
  c->out[c->out_n++] = (uint8_t)v;
}

/* processByte fsk.ts:346-375 */
static void process_byte(fsko_core *c, int bit) {
  int bitPosition = c->bit_position;
  int stopBitPosition = c->cfg.parity == 0 ? 9 : 10;
  if (bitPosition == 0) {
    if (bit != 0) { reset_state(c); return; }
    if (c->q_on && c->q_prev_d0 == 0) { c->q_f0_sum += c->q_last_post; c->q_starts += 1; } /* 0 1 -> 0: see fsk_oracle.h */
  } else if (bitPosition >= 1 && bitPosition <= 8) {
    c->byte_current |= (bit << (8 - bitPosition));
  } else if (c->cfg.parity != 0 && bitPosition == 9) {
    /* parity bit: not validated */
  } else if (bitPosition == stopBitPosition) {
    if (bit != 1) { c->started = 0; return; }
    if (c->q_on) { /* a byte completes: the stop bit's vote and the discriminator output at its decision instant */
      double ones = c->q_vote_ones, cnt = c->q_vote_count, zeros = cnt - ones;
      c->q_eye_sum += fabs(2 * ones - cnt) / cnt;
      c->q_minor += ones < zeros ? ones : zeros;
      c->q_votes += cnt;
      if ((c->byte_current & 3) == 2) { /* data ends 1 0, then the stop bit: 1 0 -> 1, the mirror image of 0 1 -> 0 */
        c->q_f_sum += c->q_last_post;
        c->q_f2_sum += c->q_last_post * c->q_last_post;
        c->q_ftrans += 1;
      }
      c->q_prev_d0 = c->byte_current & 1;
      c->q_bytes += 1;
    }
    push_byte(c, c->byte_current);
    c->byte_current = 0;
    c->bit_position = -1;
  } else {
    c->started = 0;
    return;
  }
  c->bit_position++;
}

/* processDownsampledBit fsk.ts:278-344 */
static void process_downsampled_bit(fsko_core *c, int bitValue, double amplitude) {
  ring_put_u8(&c->sync_ring, bitValue);
  ring_put_f32(&c->amp_ring, amplitude);

  c->global_sample_counter += 1;
  if (amplitude < c->silence_threshold) {
    c->silence_count += 1;
    if (c->silence_count >= c->samples_for_eod) {
      c->eod_count += 1; /* emit('eod') */
      if (c->q_on && c->q_armed) { /* first 'eod' after a sync: the amplitude floor of the silence that caused it */
        double n = floor(c->samples_for_eod), sum = 0, i;
        if (n > c->amp_ring.length) n = c->amp_ring.length;
        for (i = 0; i < n; i += 1) sum += ring_get_f32(&c->amp_ring, c->amp_ring.length - 1 - i);
        c->q_floor = n > 0 ? sum / n : 0;
        c->q_frames += 1;
        c->q_armed = 0;
      }
      reset_state(c);
      return;
    }
  } else {
    c->silence_count = 0;
  }

  if (!c->started) {
    double dsSPB = c->downsampledSamplesPerBit;
    double sampleCount = c->n_pattern * dsSPB;
    double cadence = floor(dsSPB / 4 + 0.5); /* Math.round */
    double matched = 0, total = 0;
    if (c->sync_ring.length >= sampleCount && cadence != 0 &&
        fmod(c->global_sample_counter, cadence) == 0) {
      int j;
      double k;
      for (j = 0; j < c->n_pattern; j++) {
        for (k = 0; k < dsSPB; k += 1) {
          int v = ring_get_u8(&c->sync_ring, c->sync_ring.length - (j * dsSPB + k) - 1);
          if (v == c->pattern[c->n_pattern - j]) matched += 1;
          total += 1;
        }
      }
      {
        double matchRatio = total > 0 ? matched / total : 0;
        if (matchRatio > c->cfg.syncThreshold) {
          double sum = 0, i;
          c->started = 1;
          c->byte_current = 0; c->bit_position = 0;
          c->bit_accumulator = 0; c->bit_accum_count = 0; c->bit_sample_counter = 0; c->next_bit_sample_index = 0;
          c->sync_detections += 1;
          for (i = 0; i < c->amp_ring.length; i += 1) sum += ring_get_f32(&c->amp_ring, i);
          c->silence_threshold = (sum / c->amp_ring.length) * 0.1;
          if (c->q_on) {
            const fsko_config *g = &c->cfg; /* the byte in front of the first start bit is the last one of the pattern */
            int last = g->sfdLen > 0 ? g->sfdPattern[g->sfdLen - 1] : g->preambleLen > 0 ? g->preamblePattern[g->preambleLen - 1] : 1;
            c->q_signal = sum / c->amp_ring.length; c->q_armed = 1; c->q_prev_d0 = last & 1;
          }
        }
      }
    }
  } else {
    c->bit_accumulator += bitValue;
    c->bit_accum_count += 1;
    c->bit_sample_counter += 1;
    if (c->bit_sample_counter >= c->next_bit_sample_index) {
      int bit = c->bit_accumulator > (c->bit_accum_count / 2) ? 1 : 0;
      c->q_vote_ones = c->bit_accumulator; c->q_vote_count = c->bit_accum_count;
      c->bit_accumulator = 0; c->bit_accum_count = 0;
      c->next_bit_sample_index += c->downsampledSamplesPerBit;
      process_byte(c, bit);
    }
  }
}

/* processSample fsk.ts:224-276 */
static void process_sample(fsko_core *c, double sample) {
  double omega = 2 * M_PI * c->centerFreq / c->cfg.sampleRate;
  double i = sample * cos(c->local_osc_phase);
  double q = sample * sin(c->local_osc_phase);
  c->local_osc_phase = fmod(c->local_osc_phase + omega, 2 * M_PI);
  i = fsko_iir_process(c->lp_i, i);
  q = fsko_iir_process(c->lp_q, q);
  c->ds_i += i;
  c->ds_q += q;
  c->ds_counter += 1;
  if (c->ds_counter >= c->downsampleRatio) {
    double avgI = c->ds_i / c->downsampleRatio;
    double avgQ = c->ds_q / c->downsampleRatio;
    double currentPhase = atan2(avgQ, avgI);
    double amplitude = sqrt(avgI * avgI + avgQ * avgQ);
    double phaseDiff = currentPhase - c->last_phase;
    double filtered;
    int bitValue;
    if (phaseDiff > M_PI) phaseDiff -= 2 * M_PI;
    else if (phaseDiff < -M_PI) phaseDiff += 2 * M_PI;
    c->last_phase = currentPhase;
    filtered = fsko_iir_process(c->post, phaseDiff);
    bitValue = filtered > 0 ? 1 : 0;
    c->ds_i = 0; c->ds_q = 0; c->ds_counter = 0;
    if (c->trace && c->trace->n < c->trace->cap) {
      fsko_trace *t = c->trace;
      if (t->bit) t->bit[t->n] = (uint8_t)bitValue;
      if (t->amp) t->amp[t->n] = amplitude;
      if (t->post_in) t->post_in[t->n] = phaseDiff;
      if (t->post_out) t->post_out[t->n] = filtered;
      t->n++;
    }
    c->q_last_post = filtered;
    process_downsampled_bit(c, bitValue, amplitude);
  }
}

/* AGCProcessor.process fsk.ts:52-76 (in place, Float32Array store + reload) */
static void agc_process(fsko_core *c, float *samples, size_t n) {
  size_t i;
  for (i = 0; i < n; i++) {
    double outputLevel;
    samples[i] = (float)((double)samples[i] * c->agc_gain);
    outputLevel = fabs((double)samples[i]);
    if (outputLevel > c->agc_target) {
      double targetGain = c->agc_target / outputLevel;
      c->agc_gain += (targetGain - c->agc_gain) * c->agc_attack;
    } else if (outputLevel > 0) {
      double targetGain = c->agc_target / outputLevel;
      c->agc_gain += (targetGain - c->agc_gain) * c->agc_release;
    }
    { /* Math.max(0.1, Math.min(10.0, g)); the gain cannot become NaN (NaN samples fail both compares) */
      double g = c->agc_gain < 10.0 ? c->agc_gain : 10.0;
      c->agc_gain = g > 0.1 ? g : 0.1;
    }
  }
}

/* demodulateData fsk.ts:190-222 */
long fsko_demodulate(fsko_core *c, float *samples, size_t n, uint8_t *out, size_t out_cap, uint32_t *eod_count) {
  size_t i;
  double eod0;
  long produced;
  if (!c || !c->ready) return -1; /* throws 'FSK demodulator not configured' */
  c->demod_calls += 1;
  c->total_samples += (double)n;
  eod0 = c->eod_count;
  c->out_n = 0;
  if (c->has_agc) agc_process(c, samples, n);
  /* preFilter.processBuffer -> new Float32Array, then per-sample loop.  The pre-filter has no
   * dependence on the per-sample state machine, so interleaving is equivalent. */
  for (i = 0; i < n; i++) {
    float pre = (float)fsko_iir_process(c->pre, (double)samples[i]);
    if (c->trace && c->trace->pre_out && c->trace->pre_n < c->trace->pre_cap) c->trace->pre_out[c->trace->pre_n++] = pre;
    process_sample(c, (double)pre);
  }
  produced = (long)c->out_n;
  if (out && out_cap && c->out_n) memcpy(out, c->out, c->out_n < out_cap ? c->out_n : out_cap);   /* (c->out is NULL until the first byte: memcpy from NULL is undefined even for 0 bytes -- found by the UBSan run) */
  c->out_n = 0;
  if (eod_count) *eod_count = (uint32_t)(c->eod_count - eod0);
  return produced;
}

void fsko_v8_sin(const double *x, double *y, size_t n) {
  size_t i;
  for (i = 0; i < n; i++) y[i] = v8_sin(x[i]);
}

/* generateFSKSignalInternal fsk.ts:389-424 */
long fsko_modulate_length(const fsko_core *c, size_t n_bytes) {
  double totalBytes, padding, silence;
  if (!c || !c->ready) return -1;
  totalBytes = (double)c->cfg.preambleLen + (double)c->cfg.sfdLen + (double)n_bytes;
  padding = totalBytes > 0 ? c->samplesPerBit * 2 : 0;
  silence = c->bitsPerByte * c->samplesPerBit;
  return (long)(totalBytes * c->bitsPerByte * c->samplesPerBit + padding + silence);
}

typedef struct modstate { float *out; long n, idx; double phase; const fsko_core *c; } modstate;
static void gen_bit(modstate *m, int bit) { /* fsk.ts:400-406 */
  double frequency = bit == 1 ? m->c->cfg.markFrequency : m->c->cfg.spaceFrequency;
  long i;
  for (i = 0; i < (long)m->c->samplesPerBit && m->idx < m->n; i++) {
    m->out[m->idx++] = (float)v8_sin(m->phase); /* Math.sin as V8 computes it: v8_sin.h */
    m->phase += 2 * M_PI * frequency / m->c->cfg.sampleRate;
  }
}
static void gen_byte(modstate *m, int byte) { /* fsk.ts:408-420 */
  int i;
  for (i = 0; i < m->c->cfg.startBits; i++) gen_bit(m, 0);
  for (i = 7; i >= 0; i--) gen_bit(m, (byte >> i) & 1);
  if (m->c->cfg.parity != 0) {
    int parity = 0;
    for (i = 0; i < 8; i++) parity ^= (byte >> i) & 1;
    gen_bit(m, m->c->cfg.parity == 1 ? parity : 1 - parity);
  }
  for (i = 0; i < m->c->cfg.stopBits; i++) gen_bit(m, 1);
}
long fsko_modulate(const fsko_core *c, const uint8_t *data, size_t n_bytes, float *out, size_t out_cap) {
  modstate m;
  long total = fsko_modulate_length(c, n_bytes);
  double totalBytes;
  size_t i;
  if (total < 0) return -1; /* throws 'FSK modulator not configured' */
  if ((size_t)total > out_cap) return -2;
  memset(out, 0, sizeof(float) * (size_t)total);
  totalBytes = (double)c->cfg.preambleLen + (double)c->cfg.sfdLen + (double)n_bytes;
  m.out = out; m.n = total; m.c = c; m.phase = 0;
  m.idx = totalBytes > 0 ? (long)(c->samplesPerBit * 2) : 0;
  for (i = 0; i < (size_t)c->cfg.preambleLen; i++) gen_byte(&m, c->cfg.preamblePattern[i]);
  for (i = 0; i < (size_t)c->cfg.sfdLen; i++) gen_byte(&m, c->cfg.sfdPattern[i]);
  for (i = 0; i < n_bytes; i++) gen_byte(&m, data[i]);
  return total;
}

/* reset() fsk.ts:464-469: FSKCore overrides BaseModulator.reset, `ready` stays true */
void fsko_reset(fsko_core *c) {
  reset_state(c);
  if (c->rings_ok) ring_clear(&c->sync_ring);
  c->out_n = 0;
  c->sync_detections = 0; c->demod_calls = 0; c->total_samples = 0;
}

/* getStatus() fsk.ts:481-493 */
void fsko_get_status(const fsko_core *c, fsko_status *st) {
  memset(st, 0, sizeof(*st));
  st->ready = c->ready;
  st->frameStarted = c->started;
  st->globalSampleCounter = c->global_sample_counter;
  st->receivedBitsLength = c->rings_ok ? c->sync_ring.length : 0;
  st->byteBufferLength = (double)c->out_n;
  st->demodulationCalls = c->demod_calls;
  st->syncDetections = c->sync_detections;
  st->silenceThreshold = c->silence_threshold;
  st->totalSamplesProcessed = c->total_samples;
  st->agcGain = c->has_agc ? c->agc_gain : NAN;
  st->eodCount = c->eod_count;
}

/* ---- signal-quality estimates (an extension: the reference's getSignalQuality() is an all-zero stub) ---------------- */
void fsko_enable_quality(fsko_core *c, int on) {
  c->q_on = on ? 1 : 0;
  c->q_armed = 0;
  c->q_signal = c->q_floor = c->q_frames = c->q_f_sum = c->q_f2_sum = c->q_eye_sum = c->q_bytes = c->q_minor = c->q_votes = 0;
  c->q_f0_sum = c->q_starts = c->q_ftrans = 0;
}
void fsko_get_quality(const fsko_core *c, fsko_quality *q) {
  memset(q, 0, sizeof(*q));
  q->frames = c->q_frames; q->bytes = c->q_bytes;
  q->signalLevel = c->q_signal; q->noiseFloor = c->q_floor;
  if (c->q_frames > 0) q->snr = c->q_floor > 0 ? fmin(200.0, 20 * log10(c->q_signal / c->q_floor)) : 200.0;
  if (c->q_votes > 0) q->ber = c->q_minor / c->q_votes;
  if (c->q_bytes > 0) q->eyeOpening = c->q_eye_sum / c->q_bytes;
  if (c->q_ftrans > 0) {
    double m = c->q_f_sum / c->q_ftrans, v = c->q_f2_sum / c->q_ftrans - m * m;
    q->phaseJitter = sqrt(v > 0 ? v : 0);
    if (c->q_starts > 0) /* mid-point of the two mirrored transitions: the post filter's lag cancels, a carrier offset does not.
                            The mixer turns a tone ABOVE the centre into a phase that falls (fsk.ts:229-230), hence the sign */
      q->frequencyOffset = -0.5 * (m + c->q_f0_sum / c->q_starts) * (c->cfg.sampleRate / c->downsampleRatio) / (2 * M_PI);
  }
}
