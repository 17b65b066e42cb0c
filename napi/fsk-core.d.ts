// Typings of napi/fsk-core.js: the reference's FSKConfig / IModulator surface (src/modems/fsk.ts:5-17,
// src/core.ts:88-117) plus the batch interface.
export interface FSKConfig {
  sampleRate: number; baudRate: number; markFrequency: number; spaceFrequency: number;
  preamblePattern: number[]; sfdPattern: number[]; startBits: number; stopBits: number;
  parity: 'none' | 'even' | 'odd'; syncThreshold: number; agcEnabled: boolean;
  preFilterBandwidth: number; adaptiveThreshold: boolean;
}
export const DEFAULT_FSK_CONFIG: FSKConfig;
export const PRECISION_F32: 0;
export const PRECISION_F64: 1;
export class Event { constructor(data?: unknown); readonly data: unknown; }
export class EventEmitter {
  on(eventName: string, callback: (event: Event) => void): void;
  off(eventName: string, callback: (event: Event) => void): void;
  emit(eventName: string, event?: Event): void;
  removeAllListeners(eventName?: string): void;
}
export interface FSKStatus {
  ready: boolean; frameStarted: boolean; globalSampleCounter: number; receivedBitsLength: number;
  byteBufferLength: number; demodulationCalls: number; syncDetections: number; silenceThreshold: number;
  totalSamplesProcessed: number; agcGain?: number; eodCount?: number;
}
export class FSKCore extends EventEmitter {
  constructor(options?: { device?: number; precision?: 0 | 1 });
  readonly name: 'FSK'; readonly type: 'FSK';
  configure(config: Partial<FSKConfig>): void;
  getConfig(): FSKConfig;
  modulateData(data: Uint8Array): Promise<Float32Array>;
  demodulateData(samples: Float32Array): Promise<Uint8Array>;
  reset(): void;
  isReady(): boolean;
  getSignalQuality(): { snr: number; ber: number; eyeOpening: number; phaseJitter: number; frequencyOffset: number };
  /** opt-in extension: real estimates (include/fskhip.h), off by default; getSignalQuality() keeps the reference's zeros */
  enableSignalQualityEstimates(on?: boolean): void;
  getSignalQualityEstimates(): SignalQualityEstimates;
  getStatus(): FSKStatus;
  close(): void;
}
export interface SignalQualityEstimates {
  snr: number; ber: number; eyeOpening: number; phaseJitter: number; frequencyOffset: number;
  signalLevel: number; noiseFloor: number; frames: number; bytes: number;
}
export class FSKBatch {
  constructor(nStreams: number, configs: Partial<FSKConfig> | Partial<FSKConfig>[], options?: { device?: number; precision?: 0 | 1 });
  demodulateData(samples: Float32Array, nPerStream: number, pitch?: number, writebackAgc?: boolean): { bytes: Uint8Array[]; eod: Uint32Array };
  /** the same on a libuv worker thread (N-API async work); one call in flight per batch */
  demodulateDataAsync(samples: Float32Array, nPerStream: number, pitch?: number, writebackAgc?: boolean): Promise<{ bytes: Uint8Array[]; eod: Uint32Array }>;
  modulateData(payloads: Uint8Array[]): Float32Array[];
  reset(stream?: number): void;
  getStatus(stream?: number): FSKStatus;
  /** 1 = the stream absorbed a NaN / Inf sample (dead from there on, like the reference's instance) or, fp32 engines, a sample beyond their range */
  getFaults(): Uint8Array;
  enableSignalQualityEstimates(on?: boolean): void;
  getSignalQualityEstimates(stream?: number): SignalQualityEstimates;
  close(): void;
}
/** one Node process, several GPUs: one FSKBatch per device over contiguous stream blocks, calls issued together */
export class FSKBatchSharded {
  constructor(nStreams: number, configs: Partial<FSKConfig> | Partial<FSKConfig>[], options?: { devices?: number[]; precision?: 0 | 1 });
  readonly shards: { first: number; count: number; device: number; batch: FSKBatch }[];
  demodulateData(samples: Float32Array, nPerStream: number, pitch?: number, writebackAgc?: boolean): Promise<{ bytes: Uint8Array[]; eod: Uint32Array }>;
  modulateData(payloads: Uint8Array[]): Float32Array[];
  reset(stream?: number): void;
  getStatus(stream?: number): FSKStatus;
  close(): void;
}
