'use strict';
// FSKProcessorBatch: S instances of the reference's FSKProcessor (src/webaudio/processors/fsk-processor.ts) on one
// GPU.  process(inputs, nOut) is process() for every stream in one call: demodulated bytes go into per-stream RX
// rings on the device, pending modulations are fed from the device; modulate / demodulate / reset / status mirror
// the worklet's message handlers (without the waiting: callers poll).
const path = require('path');
const addon = require(path.join(__dirname, 'fsk_addon.node'));
const PROC_CLEAR_RX_ON_TX_COMPLETE = 1, PROC_GRAPH = 2;

class FSKProcessorBatch {
  // batch: an FSKBatch (fsk-core.js); rxCapacity 1024 = demodulatedBuffer (fsk-processor.ts:84)
  constructor(batch, options = {}) {
    this.batch = batch;
    this.nStreams = batch.nStreams;
    this.rxCapacity = options.rxCapacity || 1024;
    this.flags = (options.clearRxOnTxComplete === false ? 0 : PROC_CLEAR_RX_ON_TX_COMPLETE) | (options.useGraph ? PROC_GRAPH : 0);
    this.handle = addon.processorCreate(batch.handle, this.rxCapacity);
    this.processDemodulationCallCount = 0;
  }
  close() { if (this.handle) { addon.processorDestroy(this.handle); this.handle = null; } }

  // process(inputs, outputs) fsk-processor.ts:152-167.  inputs: Float32Array [S][nIn] or null; returns Float32Array [S][nOut] or null
  process(inputs, nIn, nOut) {
    if (inputs) this.processDemodulationCallCount++;
    return addon.processorProcess(this.handle, inputs || null, inputs ? nIn : 0, inputs ? nIn : 0, nOut || 0, this.flags);
  }
  // 'modulate' (87-113): payloads = array of S Uint8Array; mask = optional array of S booleans
  modulate(payloads, mask) {
    let mx = 1;
    for (const p of payloads) mx = Math.max(mx, p.length);
    const slab = new Uint8Array(mx * this.nStreams);
    const lens = new Uint32Array(this.nStreams);
    payloads.forEach((p, i) => { slab.set(p, i * mx); lens[i] = p.length; });
    try {
      addon.processorModulate(this.handle, slab, lens, mx, mask ? Uint8Array.from(mask.map((b) => (b ? 1 : 0))) : null);
    } catch (e) {
      if (e.code === '-8') throw new Error('Modulation already in progress');   // fsk-processor.ts:91
      throw e;
    }
  }
  txState() { return addon.processorTxState(this.handle); }
  // 'demodulate' (117-138) without the wait: everything buffered, per stream
  demodulate() {
    const r = addon.processorDrain(this.handle, this.rxCapacity);
    const out = [];
    for (let s = 0; s < this.nStreams; s++) out.push(r.out.slice(s * r.outPitch, s * r.outPitch + r.counts[s]));
    return out;
  }
  rxLengths() { return addon.processorRxLength(this.handle); }
  reset(stream = -1) { addon.processorReset(this.handle, stream); }
  status(stream = 0) {               // the 'status' reply (240-253)
    const tx = this.txState();
    return Object.assign({ demodulatedBufferLength: this.rxLengths()[stream], pendingModulation: !!tx.pending[stream],
      fskCoreReady: true, processDemodulationCallCount: this.processDemodulationCallCount }, this.batch.getStatus(stream));
  }
}
module.exports = { FSKProcessorBatch, PROC_CLEAR_RX_ON_TX_COMPLETE, PROC_GRAPH };
