'use strict';
// src/dsp/filters.ts with the reference's names: FilterDesign.butterworth* / sinc*, IIRFilter, FIRFilter, FilterFactory.createIIR* /
// createFIR*, plus IIRFilterBatch / FIRFilterBatch (S streams per call).  Designs and filtering run in libfskhip.so.
const path = require('path');
const addon = require(path.join(__dirname, 'fsk_addon.node'));
const PRECISION_F32 = 0, PRECISION_F64 = 1;

function bw(kind, f, bandwidth, sampleRate) {
  const c = addon.butterworth(kind, f, bandwidth, sampleRate);
  return { b: [c[0], c[1], c[2]], a: [c[3], c[4], c[5]] };
}
class FilterDesign {
  static butterworthLowpass(cutoffFreq, sampleRate) { return bw(0, cutoffFreq, 0, sampleRate); }          // filters.ts:180-192
  static butterworthHighpass(cutoffFreq, sampleRate) { return bw(1, cutoffFreq, 0, sampleRate); }         // filters.ts:200-212
  static butterworthBandpass(centerFreq, bandwidth, sampleRate) { return bw(2, centerFreq, bandwidth, sampleRate); }   // filters.ts:221-234
  static sincLowpass(cutoffFreq, sampleRate, numTaps) { return Array.from(addon.sincLowpass(cutoffFreq, sampleRate, numTaps)); }
  static sincHighpass(cutoffFreq, sampleRate, numTaps) { return Array.from(addon.sincHighpass(cutoffFreq, sampleRate, numTaps)); }
  static sincBandpass(centerFreq, bandwidth, sampleRate, numTaps) { return Array.from(addon.sincBandpass(centerFreq, bandwidth, sampleRate, numTaps)); }
}

// [nStreams][n] in one typed array: n, or an error for a length that is not a whole number of samples per stream
function samplesPerStream(input, nStreams) {
  if (input.length % nStreams !== 0) throw new RangeError('input length ' + input.length + ' is not a multiple of the stream count ' + nStreams);
  return input.length / nStreams;
}

class FIRFilterBatch {
  constructor(coefficients, nStreams = 1, options = {}) {
    this.coefficients = [...coefficients];
    this.nStreams = nStreams;
    this.handle = addon.firCreate(Float64Array.from(coefficients), nStreams, options.device || 0,
      options.precision === undefined ? PRECISION_F64 : options.precision);
  }
  // input: Float32Array [nStreams][n]
  processBuffer(input) {
    const n = samplesPerStream(input, this.nStreams);
    if (n === 0) return new Float32Array(0);
    return addon.firProcess(this.handle, input, n, n, this.nStreams);
  }
  reset(stream = -1) { addon.firReset(this.handle, stream); }
  getCoefficients() { return [...this.coefficients]; }
  close() { if (this.handle) { addon.firDestroy(this.handle); this.handle = null; } }
}

class FIRFilter extends FIRFilterBatch {          // filters.ts:112-167
  constructor(coefficients, options = {}) { super(coefficients, 1, options); }
  process(input) { return this.processBuffer(Float32Array.of(input))[0]; }
}

class IIRFilterBatch {                            // filters.ts:8-106, S streams per call
  constructor(b, a, nStreams = 1, options = {}) {
    // the constructor's three errors (filters.ts:19-21), before anything touches the GPU
    if (!b || b.length === 0) throw new Error('Feedforward coefficients (b) cannot be empty');
    if (!a || a.length === 0) throw new Error('Feedback coefficients (a) cannot be empty');
    if (a[0] === 0) throw new Error('First feedback coefficient (a[0]) cannot be zero');
    this.nStreams = nStreams;
    this.handle = addon.iirCreate(Float64Array.from(b), Float64Array.from(a), nStreams, options.device || 0,
      options.precision === undefined ? PRECISION_F64 : options.precision);
  }
  // Float32Array [nStreams][n] -> Float32Array (processBuffer, filters.ts:81-87)
  processBuffer(input) {
    const n = samplesPerStream(input, this.nStreams);
    if (n === 0) return new Float32Array(0);
    return addon.iirProcess(this.handle, input, n, n, this.nStreams);
  }
  // Float64Array [nStreams][n] -> Float64Array: what process() returns sample by sample (nothing rounded to float)
  processSamples(input) {
    const n = samplesPerStream(input, this.nStreams);
    if (n === 0) return new Float64Array(0);
    return addon.iirProcess(this.handle, input, n, n, this.nStreams);
  }
  reset(stream = -1) { addon.iirReset(this.handle, stream); }
  getCoefficients() {
    const c = addon.iirCoefficients(this.handle), nb = c[0], na = c[1];
    return { b: Array.from(c.subarray(2, 2 + nb)), a: Array.from(c.subarray(2 + nb, 2 + nb + na)) };
  }
  close() { if (this.handle) { addon.iirDestroy(this.handle); this.handle = null; } }
}

class IIRFilter extends IIRFilterBatch {          // filters.ts:8-106
  constructor(b, a, options = {}) { super(b, a, 1, options); }
  process(input) { return this.processSamples(Float64Array.of(input))[0]; }
}

class FilterFactory {                             // filters.ts:325-368
  static createIIRLowpass(cutoffFreq, sampleRate) { const d = FilterDesign.butterworthLowpass(cutoffFreq, sampleRate); return new IIRFilter(d.b, d.a); }
  static createIIRHighpass(cutoffFreq, sampleRate) { const d = FilterDesign.butterworthHighpass(cutoffFreq, sampleRate); return new IIRFilter(d.b, d.a); }
  static createIIRBandpass(centerFreq, bandwidth, sampleRate) { const d = FilterDesign.butterworthBandpass(centerFreq, bandwidth, sampleRate); return new IIRFilter(d.b, d.a); }
  static createFIRLowpass(cutoffFreq, sampleRate, numTaps = 51) { return new FIRFilter(FilterDesign.sincLowpass(cutoffFreq, sampleRate, numTaps)); }
  static createFIRHighpass(cutoffFreq, sampleRate, numTaps = 51) { return new FIRFilter(FilterDesign.sincHighpass(cutoffFreq, sampleRate, numTaps)); }
  static createFIRBandpass(centerFreq, bandwidth, sampleRate, numTaps = 51) { return new FIRFilter(FilterDesign.sincBandpass(centerFreq, bandwidth, sampleRate, numTaps)); }
}
module.exports = { FilterDesign, IIRFilter, IIRFilterBatch, FIRFilter, FIRFilterBatch, FilterFactory, PRECISION_F32, PRECISION_F64 };
