// Typings of napi/filters.js: the FIR half of src/dsp/filters.ts.
export declare const PRECISION_F32: 0;
export declare const PRECISION_F64: 1;
export declare class FilterDesign {
  static sincLowpass(cutoffFreq: number, sampleRate: number, numTaps: number): number[];
  static sincHighpass(cutoffFreq: number, sampleRate: number, numTaps: number): number[];
  static sincBandpass(centerFreq: number, bandwidth: number, sampleRate: number, numTaps: number): number[];
}
export declare class FIRFilterBatch {
  constructor(coefficients: number[], nStreams?: number, options?: { device?: number; precision?: 0 | 1 });
  /** input: [nStreams][n] */
  processBuffer(input: Float32Array): Float32Array;
  reset(stream?: number): void;
  getCoefficients(): number[];
  close(): void;
}
export declare class FIRFilter extends FIRFilterBatch {
  constructor(coefficients: number[], options?: { device?: number; precision?: 0 | 1 });
  process(input: number): number;
}
export declare class FilterFactory {
  static createFIRLowpass(cutoffFreq: number, sampleRate: number, numTaps?: number): FIRFilter;
  static createFIRHighpass(cutoffFreq: number, sampleRate: number, numTaps?: number): FIRFilter;
  static createFIRBandpass(centerFreq: number, bandwidth: number, sampleRate: number, numTaps?: number): FIRFilter;
}
