/*
 * fskhip_next.h -- C ABI of libfskhip.so for the rows either side of the demodulator hot path
 * (SURVEY.md 8(f)): the FSKProcessor streaming contract (RX byte ring, ChunkedModulator slice feeder,
 * one process() per 128-sample quantum), CRC-16-CCITT + XModem packet build/validation over the
 * demodulated bytes, and the FIR half of dsp/filters.ts.  Same conventions as fskhip.h: plain pointers
 * and sizes, FSKHIP_OK or a negative FSKHIP_E_* code, fskhip_last_error() for the text, no CPU fallback.
 */
#ifndef FSKHIP_NEXT_H
#define FSKHIP_NEXT_H

#include "fskhip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define FSKHIP_E_BUSY (-8) /* reference: throws 'Modulation already in progress' fsk-processor.ts:90-92 */

/* ---------------------------------------------------------------------------------------------------
 * CRC-16-CCITT and XModem packets (src/utils/crc16.ts, src/transports/xmodem/packet.ts, types.ts,
 * the receive checks of xmodem.ts:233-320) -- batch, one row per stream / packet.
 * The _device forms take device pointers, run on the CURRENT HIP device, asynchronously on `hip_stream`;
 * the _host forms take host pointers, run on `device` and are synchronous.
 * ------------------------------------------------------------------------------------------------- */

/* CRC16.calculate(data) (crc16.ts:21-38: poly 0x1021, init 0xFFFF, no final xor, MSB first) of
 * data[r*pitch .. + lens[r]) for every row r. */
int fskhip_crc16_device(const uint8_t *d_data, size_t pitch, const uint32_t *d_lens, uint32_t n_rows,
                        uint16_t *d_crc, void *hip_stream);
int fskhip_crc16_host(int device, const uint8_t *data, size_t pitch, const uint32_t *lens, uint32_t n_rows,
                      uint16_t *crc);

/* XModemPacket.serialize(XModemPacket.createData(seq, payload)) (packet.ts:21-54) per row:
 * SOH | seq | ~seq | len | payload | crc_hi | crc_lo into out[r*out_pitch ...], out_lens[r] = len + 6.
 * createData's argument checks (sequence 1-255, payload <= 255 bytes) are reported per row as out_lens[r] = 0;
 * the _host form returns FSKHIP_E_INVALID with the reference's message for the first such row. */
int fskhip_xmodem_serialize_device(const uint8_t *d_payloads, size_t payload_pitch, const uint32_t *d_lens,
                                   const uint32_t *d_seqs, uint32_t n_rows, uint8_t *d_out, size_t out_pitch,
                                   uint32_t *d_out_lens, void *hip_stream);
int fskhip_xmodem_serialize_host(int device, const uint8_t *payloads, size_t payload_pitch, const uint32_t *lens,
                                 const uint32_t *seqs, uint32_t n_rows, uint8_t *out, size_t out_pitch,
                                 uint32_t *out_lens);

/* How a scan of one burst ended. */
enum {
  FSKHIP_XM_NEED_MORE = 0,           /* ran out of bytes between packets (the reference would wait / time out) */
  FSKHIP_XM_EOT = 1,                 /* EOT seen where a packet could start (xmodem.ts:241-244) */
  FSKHIP_XM_TRUNCATED = 2,           /* ran out of bytes inside a packet (the reference's waitForBytes times out) */
  FSKHIP_XM_INVALID_SEQUENCE = 3,    /* (seq + nseq) != 255            'Invalid sequence number' xmodem.ts:270-274 */
  FSKHIP_XM_INVALID_CRC = 4,         /* CRC16(payload) != received CRC 'Invalid CRC'             xmodem.ts:287-291 */
  FSKHIP_XM_UNEXPECTED_SEQUENCE = 5  /* neither expected nor previous  'Unexpected sequence number' xmodem.ts:315-319 */
};

typedef struct fskhip_xmodem_result {
  uint32_t status;         /* FSKHIP_XM_* */
  uint32_t expected_after; /* receive.expectedSequence after the scan (xmodem.ts:303) */
  uint32_t packets;        /* statistics.packetsReceived: packets with the expected sequence whose payload + CRC arrived
                              (counted before the CRC check, xmodem.ts:280, so a bad-CRC packet is in it) */
  uint32_t dropped;        /* statistics.packetsDropped increments: bad seq pair / CRC / duplicate / unexpected */
  uint32_t consumed;       /* bytes taken out of the receive buffer (xmodem.ts:475-499): everything up to the end of the last
                              complete step; inside a truncated packet SOH (+ the 3 header bytes once complete) */
  uint32_t data_len;       /* bytes of assembled payload written (true size, even beyond data_pitch) */
  int32_t err_seq;         /* header of the packet that ended the scan with an error / truncation, else -1 */
  int32_t err_len;
  int32_t crc_rx;          /* FSKHIP_XM_INVALID_CRC: received and computed CRC, else -1 */
  int32_t crc_calc;
} fskhip_xmodem_result;

/*
 * XModemTransport's receive grammar (receiveAllPackets / receiveAndProcessPacket, xmodem.ts:233-320) over a
 * recorded burst per stream -- bytes[s*pitch .. + counts[s]) as the demodulator returned them: bytes other than
 * SOH/EOT between packets are ignored; SOH seq nseq len payload crc16 is accepted when seq is the expected
 * sequence and the CRC matches (payload appended to data[s*data_pitch ...], assembleData 322-333; expected
 * advances 1..255,1..), consumed-and-dropped when seq is the previous sequence (duplicate), and ends the scan
 * with an error status otherwise -- where the reference throws, NAKs and clears its buffer.
 * expected[s] is read as the starting expectedSequence (1-255) and left untouched; results[s].expected_after
 * carries the new value.
 */
int fskhip_xmodem_scan_device(const uint8_t *d_bytes, size_t pitch, const uint32_t *d_counts,
                              const uint32_t *d_expected, uint32_t n_streams, uint8_t *d_data, size_t data_pitch,
                              fskhip_xmodem_result *d_results, void *hip_stream);
int fskhip_xmodem_scan_host(int device, const uint8_t *bytes, size_t pitch, const uint32_t *counts,
                            const uint32_t *expected, uint32_t n_streams, uint8_t *data, size_t data_pitch,
                            fskhip_xmodem_result *results);

/* ---------------------------------------------------------------------------------------------------
 * FSKProcessor (src/webaudio/processors/fsk-processor.ts) + ChunkedModulator (src/webaudio/
 * chunked-modulator.ts), one instance per stream of an engine, state resident on the device.
 * ------------------------------------------------------------------------------------------------- */
typedef struct fskhip_processor fskhip_processor;

/* new FSKProcessor() per stream: demodulatedBuffer = RingBuffer(Uint8Array, rx_capacity) (1024 in the reference,
 * fsk-processor.ts:84), no pending modulation.  The engine must outlive the processor. */
int fskhip_processor_create(fskhip_engine *e, uint32_t rx_capacity, fskhip_processor **out);
int fskhip_processor_destroy(fskhip_processor *p);

/*
 * process(inputs, outputs) for every stream (fsk-processor.ts:152-167): demodulateFrom(input) --
 * fskCore.demodulateData(input) and every returned byte put into the stream's RX ring, overwriting the oldest
 * when full (294-322, utils.ts:38-48) -- then modulateTo(output): zero fill, the next n_out samples of the
 * pending signal, and on completion the modulation is dropped (256-276).  d_in is [n_streams][in_pitch] float32
 * with n_in samples per stream, d_out [n_streams][out_pitch] with n_out; either may be NULL (that half is
 * skipped, like a missing input/output).  FSKHIP_PROC_CLEAR_RX_ON_TX_COMPLETE also clears the stream's RX ring
 * when its modulation completes, which the reference's 'modulate' message handler does to avoid
 * self-reception (228-235).  FSKHIP_PROC_GRAPH replays the launches of a quantum as one captured hipGraph.
 * Asynchronous on `hip_stream`.
 */
#define FSKHIP_PROC_CLEAR_RX_ON_TX_COMPLETE 1u
#define FSKHIP_PROC_GRAPH 2u
int fskhip_processor_process_device(fskhip_processor *p, float *d_in, size_t n_in, size_t in_pitch, float *d_out,
                                    size_t n_out, size_t out_pitch, uint32_t flags, void *hip_stream);
int fskhip_processor_process_host(fskhip_processor *p, float *in, size_t n_in, size_t in_pitch, float *out,
                                  size_t n_out, size_t out_pitch, uint32_t flags);

/*
 * The 'modulate' request (fsk-processor.ts:87-113) for the streams with mask[s] != 0 (mask NULL = all):
 * pendingModulation = new ChunkedModulator(fskCore); startModulation(payload) (chunked-modulator.ts:31-39:
 * the whole signal is generated now; an EMPTY payload leaves the stream with a pending modulator that never
 * produces samples and never completes, as in the reference).  FSKHIP_E_BUSY ('Modulation already in progress')
 * if any selected stream still has one; nothing is started then.
 */
int fskhip_processor_modulate_host(fskhip_processor *p, const uint8_t *payloads, const uint32_t *lens,
                                   size_t payload_pitch, const uint8_t *mask);
/* ChunkedModulator state per stream: pos/total samples (isModulating() = total > 0, getProgress() = pos/total),
 * pending[s] = pendingModulation != null, completed[s] = modulations completed since create.  Any may be NULL. */
int fskhip_processor_tx_state_host(fskhip_processor *p, uint32_t *pos, uint32_t *total, uint8_t *pending,
                                   uint32_t *completed);
/* the 'demodulate' request without the wait (fsk-processor.ts:117-138): remove everything buffered.  counts[s]
 * is the number of bytes removed into out[s*out_pitch ...] (out_pitch >= rx_capacity never overflows). */
int fskhip_processor_rx_drain_host(fskhip_processor *p, uint8_t *out, size_t out_pitch, uint32_t *counts);
/* demodulatedBufferLength of the 'status' reply (fsk-processor.ts:246). */
int fskhip_processor_rx_length_host(fskhip_processor *p, uint32_t *lengths);
/* reset() (fsk-processor.ts:140-146): RX ring cleared, pending modulation dropped; stream < 0 = all.  The
 * FSKCore state is NOT reset (the reference does not either). */
int fskhip_processor_reset(fskhip_processor *p, int64_t stream);

/* ---------------------------------------------------------------------------------------------------
 * FIR half of src/dsp/filters.ts: FIRFilter (112-167) batched over streams, and the windowed-sinc designs
 * (243-314) + FilterFactory.createFIR* (346-368).
 * ------------------------------------------------------------------------------------------------- */
typedef struct fskhip_fir fskhip_fir;

/* FilterDesign.sincLowpass / sincHighpass / sincBandpass (filters.ts:243-314).  `taps` must hold n_taps + 1
 * doubles (an even n_taps is bumped to the next odd number by sincLowpass, filters.ts:244-246); the number of
 * coefficients written is returned (negative FSKHIP_E_* on error).  Host arithmetic in doubles, like the
 * reference; agreement with V8 is to the last ulp of libm's sin/cos. */
int fskhip_sinc_lowpass(double cutoff, double sampleRate, uint32_t n_taps, double *taps);
int fskhip_sinc_highpass(double cutoff, double sampleRate, uint32_t n_taps, double *taps);
int fskhip_sinc_bandpass(double center, double bandwidth, double sampleRate, uint32_t n_taps, double *taps);

/* new FIRFilter(coefficients) for n_streams independent streams sharing one coefficient set (filters.ts:117-120);
 * delay lines start at zero.  precision: FSKHIP_PRECISION_F64 accumulates output += c[i]*delay[i] in doubles in
 * the reference's order (bit-identical Float32Array output), FSKHIP_PRECISION_F32 uses fp32 FMAs. */
int fskhip_fir_create(int device, const double *taps, uint32_t n_taps, uint32_t n_streams, int precision,
                      fskhip_fir **out);
int fskhip_fir_destroy(fskhip_fir *f);
uint32_t fskhip_fir_streams(const fskhip_fir *f);   /* the n_streams it was created for (0 for NULL); ABI 7 */
/* processBuffer(input) (filters.ts:142-148) for every stream: out[s][t] = f32(sum_i c[i] * x_s[t-i]), the delay
 * line carried across calls.  in/out are [n_streams][pitch] float32 and may not alias. */
int fskhip_fir_process_device(fskhip_fir *f, const float *d_in, size_t n_per_stream, size_t in_pitch, float *d_out,
                              size_t out_pitch, void *hip_stream);
int fskhip_fir_process_host(fskhip_fir *f, const float *in, size_t n_per_stream, size_t in_pitch, float *out,
                            size_t out_pitch);
/* reset() (filters.ts:153-156) for one stream, or all when stream < 0. */
int fskhip_fir_reset(fskhip_fir *f, int64_t stream);

/* ---------------------------------------------------------------------------------------------------
 * IIR half of src/dsp/filters.ts: IIRFilter (8-106) batched over streams, + FilterFactory.createIIR* (325-344; the
 * designs themselves are fskhip_butterworth_* in fskhip.h).  Inside the demodulator the same filter lives as three
 * hard-wired biquads; this is the generic class: Direct Form I of any order <= 8.
 * ------------------------------------------------------------------------------------------------- */
typedef struct fskhip_iir fskhip_iir;

/* new IIRFilter(b, a) (filters.ts:17-42) for n_streams independent streams sharing one coefficient set; histories start
 * at zero.  The constructor's three errors come back as FSKHIP_E_INVALID with the reference's messages ('Feedforward
 * coefficients (b) cannot be empty', 'Feedback coefficients (a) cannot be empty', 'First feedback coefficient (a[0])
 * cannot be zero'); coefficients are normalised by a[0] exactly as filters.ts:30-39 does (b[i] /= a0, a[i] /= a0 for
 * i >= 1).  More than 9 coefficients on either side: FSKHIP_E_UNSUPPORTED (the kernel keeps eight past inputs and outputs
 * in registers).  precision: FSKHIP_PRECISION_F64 evaluates output += b[i] * x[n-i], output -= a[i] * y[n-i] in doubles
 * in the reference's order, every product and sum rounded on its own (bit-identical results), FSKHIP_PRECISION_F32 the same
 * operations in floats (separate multiplies and adds: the library is built with -ffp-contract=off). */
int fskhip_iir_create(int device, const double *b, uint32_t nb, const double *a, uint32_t na, uint32_t n_streams,
                      int precision, fskhip_iir **out);
int fskhip_iir_destroy(fskhip_iir *f);
uint32_t fskhip_iir_streams(const fskhip_iir *f);   /* the n_streams it was created for (0 for NULL); ABI 7 */
/* getCoefficients() (filters.ts:103-105): the normalised sets; b and a must hold 9 doubles each. */
int fskhip_iir_get_coefficients(const fskhip_iir *f, double *b, uint32_t *nb, double *a, uint32_t *na);
/* processBuffer(input) (filters.ts:81-87) for every stream: out[s][t] = f32(process(in[s][t])), the histories carried
 * across calls.  in/out are [n_streams][pitch] float32; in == out (in place) is allowed. */
int fskhip_iir_process_device(fskhip_iir *f, const float *d_in, size_t n_per_stream, size_t in_pitch, float *d_out,
                              size_t out_pitch, void *hip_stream);
int fskhip_iir_process_host(fskhip_iir *f, const float *in, size_t n_per_stream, size_t in_pitch, float *out,
                            size_t out_pitch);
/* process(input) (filters.ts:47-76) sample by sample, numbers in and out (doubles, nothing rounded to float);
 * shares the histories with the Float32Array calls above. */
int fskhip_iir_process_f64_device(fskhip_iir *f, const double *d_in, size_t n_per_stream, size_t in_pitch, double *d_out,
                                  size_t out_pitch, void *hip_stream);
int fskhip_iir_process_f64_host(fskhip_iir *f, const double *in, size_t n_per_stream, size_t in_pitch, double *out,
                                size_t out_pitch);
/* reset() (filters.ts:92-98) for one stream, or all when stream < 0. */
int fskhip_iir_reset(fskhip_iir *f, int64_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FSKHIP_NEXT_H */
