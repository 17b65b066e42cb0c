// Typings of napi/fsk-processor.js and napi/chunked-modulator.js (src/webaudio/processors/fsk-processor.ts,
// src/webaudio/chunked-modulator.ts on the device).
import { FSKBatch, FSKCore, FSKStatus } from './fsk-core';
export declare const PROC_CLEAR_RX_ON_TX_COMPLETE: 1;
export declare const PROC_GRAPH: 2;
export declare class FSKProcessorBatch {
  constructor(batch: FSKBatch, options?: { rxCapacity?: number; clearRxOnTxComplete?: boolean; useGraph?: boolean });
  readonly nStreams: number;
  /** process(inputs, outputs) for every stream: inputs [S][nIn] or null; returns [S][nOut] or null */
  process(inputs: Float32Array | null, nIn: number, nOut: number): Float32Array | null;
  /** 'modulate': throws 'Modulation already in progress' when a selected stream still has one */
  modulate(payloads: Uint8Array[], mask?: boolean[]): void;
  txState(): { pos: Uint32Array; total: Uint32Array; pending: Uint8Array; completed: Uint32Array };
  /** 'demodulate' without the wait: everything buffered, per stream */
  demodulate(): Uint8Array[];
  rxLengths(): Uint32Array;
  reset(stream?: number): void;
  status(stream?: number): FSKStatus & { demodulatedBufferLength: number; pendingModulation: boolean; fskCoreReady: boolean; processDemodulationCallCount: number };
  close(): void;
}
export interface ChunkResult { signal: Float32Array; isComplete: boolean; samplesConsumed: number; totalSamples: number; }
export declare class ChunkedModulator {
  constructor(modulator: FSKCore);
  startModulation(data: Uint8Array): Promise<void>;
  getNextSamples(sampleCount: number): ChunkResult | null;
  isModulating(): boolean;
  getProgress(): number;
  cancel(): void;
  close(): void;
}
