'use strict';
// ChunkedModulator with the surface of the reference class (src/webaudio/chunked-modulator.ts:22-88:
// startModulation / getNextSamples / isModulating / getProgress / cancel), built the other way round.
//
// The reference renders the whole signal up front and then slices a Float32Array.  Here nothing is rendered ahead:
// the pending modulation lives on the GPU as the modulator's generator state (payload bytes, f64 phase, sample
// position -- a one-stream fskhip_processor, include/fskhip_next.h), and every getNextSamples(n) asks the device for
// exactly the next n samples.  The slices are bit-identical to slices of modulateData()'s output because that
// output is itself one sequential phase accumulation (fsk.ts:398-406); tests/js/next_rows_test.js checks them
// against the steps recorded from the real reference class.
const path = require('path');
const addon = require(path.join(__dirname, 'fsk_addon.node'));

class ChunkedModulator {
  // modulator: an FSKCore of fsk-core.js (anything exposing the engine `handle` the addon created)
  constructor(modulator) {
    this.modulator = modulator;
    this.device = null;        // one-stream processor on the modulator's engine
    this.boundTo = null;       // the engine handle `device` belongs to (configure() replaces engines)
    this.total = 0;            // pendingSignal.length of the reference, 0 = nothing pending
    this.position = 0;         // samplePosition
  }

  _processor() {
    const engine = this.modulator.handle;
    if (!engine) throw new Error('FSK modulator not configured');   // what modulateData() would reject with
    if (this.boundTo !== engine) {
      if (this.device) addon.processorDestroy(this.device);
      this.device = addon.processorCreate(engine, 1);
      this.boundTo = engine;
    }
    return this.device;
  }

  async startModulation(data) {
    const bytes = data instanceof Uint8Array ? data : Uint8Array.from(data);
    this._drop();
    if (bytes.length === 0) return;                                   // empty data: nothing pending (ts:32-35)
    const proc = this._processor();
    addon.processorModulate(proc, bytes, Uint32Array.of(bytes.length), bytes.length, null);
    this.total = addon.processorTxState(proc).total[0];
    this.position = 0;
  }

  getNextSamples(sampleCount) {
    if (this.total === 0) return null;
    const take = Math.min(sampleCount, this.total - this.position);
    if (take <= 0) return null;
    // the device zero-fills past the end of the signal and completes the modulation by itself
    const signal = addon.processorProcess(this._processor(), null, 0, 0, take, 0);
    this.position += take;
    const totalSamples = this.total;
    if (this.position >= totalSamples) {
      this.total = 0;
      this.position = 0;
      return { signal, isComplete: true, samplesConsumed: totalSamples, totalSamples };
    }
    return { signal, isComplete: false, samplesConsumed: this.position, totalSamples };
  }

  isModulating() { return this.total !== 0; }
  getProgress() { return this.total !== 0 ? this.position / this.total : 0; }
  cancel() { this._drop(); }

  _drop() {
    if (this.device && this.boundTo === this.modulator.handle && this.total !== 0) addon.processorReset(this.device, 0);
    this.total = 0;
    this.position = 0;
  }

  close() {
    if (this.device) { addon.processorDestroy(this.device); this.device = null; this.boundTo = null; }
  }
}
module.exports = { ChunkedModulator };
