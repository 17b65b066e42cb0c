'use strict';
// FIR half of src/dsp/filters.ts with the reference's names: FilterDesign.sinc*, FIRFilter, FilterFactory.createFIR*,
// plus FIRFilterBatch (S streams per call).  Designs and filtering run in libfskhip.so.
const path = require('path');
const addon = require(path.join(__dirname, 'fsk_addon.node'));
const PRECISION_F32 = 0, PRECISION_F64 = 1;

class FilterDesign {
  static sincLowpass(cutoffFreq, sampleRate, numTaps) { return Array.from(addon.sincLowpass(cutoffFreq, sampleRate, numTaps)); }
  static sincHighpass(cutoffFreq, sampleRate, numTaps) { return Array.from(addon.sincHighpass(cutoffFreq, sampleRate, numTaps)); }
  static sincBandpass(centerFreq, bandwidth, sampleRate, numTaps) { return Array.from(addon.sincBandpass(centerFreq, bandwidth, sampleRate, numTaps)); }
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
    const n = input.length / this.nStreams;
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

class FilterFactory {                             // filters.ts:346-368
  static createFIRLowpass(cutoffFreq, sampleRate, numTaps = 51) { return new FIRFilter(FilterDesign.sincLowpass(cutoffFreq, sampleRate, numTaps)); }
  static createFIRHighpass(cutoffFreq, sampleRate, numTaps = 51) { return new FIRFilter(FilterDesign.sincHighpass(cutoffFreq, sampleRate, numTaps)); }
  static createFIRBandpass(centerFreq, bandwidth, sampleRate, numTaps = 51) { return new FIRFilter(FilterDesign.sincBandpass(centerFreq, bandwidth, sampleRate, numTaps)); }
}
module.exports = { FilterDesign, FIRFilter, FIRFilterBatch, FilterFactory, PRECISION_F32, PRECISION_F64 };
