'use strict';
// FSKCore / FSKBatch: the host side of the MI355X engine in the reference's own language.
//
// FSKCore keeps the public surface of the reference class (src/modems/fsk.ts:82-494 on top of
// src/core.ts:210-289): name, type, configure, getConfig, modulateData, demodulateData, reset,
// isReady, getSignalQuality, getStatus, on/off/emit with the 'configured' | 'eod' | 'error' events.
// Every DSP call goes through the N-API addon into libfskhip.so; there is no JavaScript DSP here.
const path = require('path');
const addon = require(path.join(__dirname, 'fsk_addon.node'));

const PRECISION_F32 = 0, PRECISION_F64 = 1;
const DEMOD_WRITEBACK_AGC = 1;

// DEFAULT_FSK_CONFIG (fsk.ts:19-33)
const DEFAULT_FSK_CONFIG = {
  sampleRate: 48000, baudRate: 1200, markFrequency: 1650, spaceFrequency: 1850,
  preamblePattern: [0x55, 0x55], sfdPattern: [0x7E], startBits: 1, stopBits: 1, parity: 'none',
  syncThreshold: 0.85, agcEnabled: true, preFilterBandwidth: 800, adaptiveThreshold: true
};

class Event {                       // core.ts:205-207
  constructor(data = null) { this.data = data; }
}

class EventEmitter {                // core.ts:210-244
  constructor() { this.listeners = new Map(); }
  on(eventName, callback) {
    if (!this.listeners.has(eventName)) this.listeners.set(eventName, []);
    this.listeners.get(eventName).push(callback);
  }
  off(eventName, callback) {
    const l = this.listeners.get(eventName);
    if (l) { const i = l.indexOf(callback); if (i !== -1) l.splice(i, 1); }
  }
  emit(eventName, event = new Event()) {
    const l = this.listeners.get(eventName);
    if (l) l.slice().forEach((cb) => cb(event));
  }
  removeAllListeners(eventName) {
    if (eventName) this.listeners.delete(eventName); else this.listeners.clear();
  }
}

class FSKCore extends EventEmitter {
  // precision defaults to the fp64 parity path: one stream cannot fill the GPU anyway and fp64 is
  // op-for-op with the reference's arithmetic.  FSKBatch is the throughput interface.
  constructor(options = {}) {
    super();
    this.name = 'FSK';
    this.type = 'FSK';
    this.device = options.device || 0;
    this.precision = options.precision === undefined ? PRECISION_F64 : options.precision;
    this.handle = null;
    this.config = undefined;
    this.ready = false;
  }

  configure(config) {               // fsk.ts:133-157
    const old = this.handle;
    this.config = Object.assign({}, DEFAULT_FSK_CONFIG, config);
    this.handle = addon.create(this.config, 1, this.device, this.precision);
    if (old) {
      // the reference rebuilds in place and keeps silence.threshold and the debug counters (fsk.ts:133-157)
      addon.carryOver(this.handle, old);
      addon.destroy(old);
    }
    this.ready = true;
    this.emit('configured');
  }

  getConfig() { return Object.assign({}, this.config); }
  isReady() { return this.ready; }

  async demodulateData(samples) {   // fsk.ts:190-222; mutates `samples` when AGC is on (fsk.ts:55)
    if (!this.ready || !this.config) throw new Error('FSK demodulator not configured');
    try {
      const r = addon.demodulate(this.handle, samples, samples.length, samples.length,
                                 this.config.agcEnabled ? DEMOD_WRITEBACK_AGC : 0);
      for (let i = 0; i < r.eod[0]; i++) this.emit('eod');
      return r.out.slice(0, r.counts[0]);
    } catch (error) {               // fsk.ts:218-221
      this.emit('error', new Event({ data: error }));
      return new Uint8Array(0);
    }
  }

  async modulateData(data) {        // fsk.ts:377-383
    if (!this.ready || !this.config) throw new Error('FSK modulator not configured');
    const bytes = data instanceof Uint8Array ? data : Uint8Array.from(data);
    const r = addon.modulate(this.handle, bytes.length ? bytes : new Uint8Array(1), Uint32Array.of(bytes.length),
                             Math.max(1, bytes.length));
    return r.out.slice(0, r.lens[0]);
  }

  reset() {                         // fsk.ts:464-469: `ready` stays true
    if (this.handle) addon.reset(this.handle, 0);
  }

  getSignalQuality() {              // fsk.ts:471-479: all-zero stub in the reference
    return { snr: 0, ber: 0, eyeOpening: 0, phaseJitter: 0, frequencyOffset: 0 };
  }
  // opt-in extension (include/fskhip.h): real estimates of the same five fields plus what they are made of
  enableSignalQualityEstimates(on) {
    if (!this.ready || !this.handle) throw new Error('FSK demodulator not configured');
    addon.enableSignalQuality(this.handle, on === undefined ? true : !!on);
  }
  getSignalQualityEstimates() {
    if (!this.handle) return Object.assign(this.getSignalQuality(), { signalLevel: 0, noiseFloor: 0, frames: 0, bytes: 0 });
    return addon.getSignalQualityEstimates(this.handle, 0);
  }

  getStatus() {                     // fsk.ts:481-493
    if (!this.handle) {
      return { ready: this.ready, frameStarted: false, globalSampleCounter: 0, receivedBitsLength: 0,
               byteBufferLength: 0, demodulationCalls: 0, syncDetections: 0, silenceThreshold: 0.01,
               totalSamplesProcessed: 0 };
    }
    const st = addon.getStatus(this.handle, 0);
    st.ready = this.ready;
    return st;
  }

  close() {
    if (this.handle) { addon.destroy(this.handle); this.handle = null; }
    this.ready = false;
  }
}

// S independent FSKCore instances in one engine: samples are [S][N] stream-major Float32Array.
class FSKBatch {
  constructor(nStreams, configs, options = {}) {
    this.nStreams = nStreams;
    this.configs = Array.isArray(configs) ? configs.map((c) => Object.assign({}, DEFAULT_FSK_CONFIG, c))
                                          : Object.assign({}, DEFAULT_FSK_CONFIG, configs);
    this.handle = addon.create(this.configs, nStreams, options.device || 0,
                               options.precision === undefined ? PRECISION_F32 : options.precision);
  }
  // returns {bytes: Uint8Array[S], eod: Uint32Array(S)}
  demodulateData(samples, nPerStream, pitch, writebackAgc) {
    const r = addon.demodulate(this.handle, samples, nPerStream, pitch || nPerStream, writebackAgc ? DEMOD_WRITEBACK_AGC : 0);
    const bytes = [];
    for (let s = 0; s < this.nStreams; s++) bytes.push(r.out.slice(s * r.outPitch, s * r.outPitch + r.counts[s]));
    return { bytes, eod: r.eod };
  }
  // the same off the event loop (N-API async work): the promise resolves with {bytes, eod}; `samples` must not be
  // touched until it settles, and one call may be in flight per batch
  async demodulateDataAsync(samples, nPerStream, pitch, writebackAgc) {
    const r = await addon.demodulateAsync(this.handle, samples, nPerStream, pitch || nPerStream, writebackAgc ? DEMOD_WRITEBACK_AGC : 0);
    const bytes = [];
    for (let s = 0; s < this.nStreams; s++) bytes.push(r.out.slice(s * r.outPitch, s * r.outPitch + r.counts[s]));
    return { bytes, eod: r.eod };
  }
  // payloads: Uint8Array[S] -> Float32Array[S]
  modulateData(payloads) {
    let pitch = 1;
    payloads.forEach((p) => { pitch = Math.max(pitch, p.length); });
    const flat = new Uint8Array(pitch * this.nStreams);
    const lens = new Uint32Array(this.nStreams);
    payloads.forEach((p, s) => { flat.set(p, s * pitch); lens[s] = p.length; });
    const r = addon.modulate(this.handle, flat, lens, pitch);
    const out = [];
    for (let s = 0; s < this.nStreams; s++) out.push(r.out.slice(s * r.outPitch, s * r.outPitch + r.lens[s]));
    return out;
  }
  reset(stream) { addon.reset(this.handle, stream === undefined ? -1 : stream); }
  getStatus(stream) { return addon.getStatus(this.handle, stream || 0); }
  // Uint8Array[nStreams]: 1 = the stream absorbed a NaN / Inf sample (the reference's instance is dead from there on too, and the engine
  // does what it does, bit for bit) or, fp32 engines only, a sample beyond their range (include/fskhip.h, fskhip_get_faults)
  getFaults() { return addon.getFaults(this.handle, this.nStreams); }
  enableSignalQualityEstimates(on) { addon.enableSignalQuality(this.handle, on === undefined ? true : !!on); }
  getSignalQualityEstimates(stream) { return addon.getSignalQualityEstimates(this.handle, stream || 0); }
  close() { if (this.handle) { addon.destroy(this.handle); this.handle = null; } }
}

// One Node process, several GPUs: one FSKBatch per device, each owning a contiguous block of streams (the layout of
// webaudio_modem_amd/sharding.py: sizes differ by at most one), the per-device calls issued together as N-API async work
// (libuv pool threads; every libfskhip entry point selects its engine's device).  No collective: streams are independent.
class FSKBatchSharded {
  // options.devices: device ordinals (default: all of addon.deviceCount()); options.precision as FSKBatch
  constructor(nStreams, configs, options = {}) {
    let devices = options.devices;
    if (!devices) {
      const n = addon.deviceCount();
      devices = [];
      for (let d = 0; d < Math.max(n, 1); d++) devices.push(d);   // n == 0: FSKBatch below fails loudly (no CPU path)
    }
    if (Array.isArray(configs) && configs.length !== nStreams) throw new Error('need one config per stream');
    this.nStreams = nStreams;
    this.shards = [];
    const base = Math.floor(nStreams / devices.length), extra = nStreams % devices.length;
    let first = 0;
    try {
      devices.forEach((device, r) => {
        const count = base + (r < extra ? 1 : 0);
        if (count > 0) {
          const cfg = Array.isArray(configs) ? configs.slice(first, first + count) : configs;
          this.shards.push({ first, count, device, batch: new FSKBatch(count, cfg, Object.assign({}, options, { device })) });
        }
        first += count;
      });
    } catch (e) {
      this.close();
      throw e;
    }
  }
  locate(stream) {
    for (const sh of this.shards) if (stream >= sh.first && stream < sh.first + sh.count) return [sh, stream - sh.first];
    throw new Error('stream out of range');
  }
  // samples: Float32Array [S][pitch]; resolves with {bytes: Uint8Array[S], eod: Uint32Array(S)} in stream order
  async demodulateData(samples, nPerStream, pitch, writebackAgc) {
    const p = pitch || nPerStream;
    const parts = await Promise.all(this.shards.map((sh) =>
      sh.batch.demodulateDataAsync(samples.subarray(sh.first * p, (sh.first + sh.count) * p), nPerStream, p, writebackAgc)));
    const bytes = [];
    const eod = new Uint32Array(this.nStreams);
    parts.forEach((r, i) => { r.bytes.forEach((b) => bytes.push(b)); eod.set(r.eod, this.shards[i].first); });
    return { bytes, eod };
  }
  modulateData(payloads) {
    if (payloads.length !== this.nStreams) throw new Error('need one payload per stream');
    let out = [];
    for (const sh of this.shards) out = out.concat(sh.batch.modulateData(payloads.slice(sh.first, sh.first + sh.count)));
    return out;
  }
  reset(stream) {
    if (stream === undefined || stream < 0) this.shards.forEach((sh) => sh.batch.reset());
    else { const [sh, local] = this.locate(stream); sh.batch.reset(local); }
  }
  getStatus(stream) { const [sh, local] = this.locate(stream || 0); return sh.batch.getStatus(local); }
  close() { this.shards.forEach((sh) => sh.batch.close()); this.shards = []; }
}

module.exports = { FSKCore, FSKBatch, FSKBatchSharded, DEFAULT_FSK_CONFIG, Event, EventEmitter, PRECISION_F32, PRECISION_F64, addon };
