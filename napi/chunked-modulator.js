'use strict';
// ChunkedModulator (src/webaudio/chunked-modulator.ts:22-88), same surface, over any IModulator -- in practice the
// FSKCore of fsk-core.js, whose modulateData runs on the GPU.  Host logic only: slicing a Float32Array.
class ChunkedModulator {
  constructor(modulator) {
    this.modulator = modulator;
    this.pendingSignal = null;
    this.samplePosition = 0;
  }
  async startModulation(data) {
    if (!data.length) { this.reset(); return; }
    this.pendingSignal = await this.modulator.modulateData(data);
    this.samplePosition = 0;
  }
  getNextSamples(sampleCount) {
    if (!this.pendingSignal) return null;
    const remaining = this.pendingSignal.length - this.samplePosition;
    if (remaining <= 0) return null;
    const samplesCount = Math.min(sampleCount, remaining);
    const signal = this.pendingSignal.slice(this.samplePosition, this.samplePosition + samplesCount);
    this.samplePosition += samplesCount;
    if (this.samplePosition >= this.pendingSignal.length) {
      const totalSamples = this.pendingSignal.length;
      this.reset();
      return { signal, isComplete: true, samplesConsumed: totalSamples, totalSamples };
    }
    return { signal, isComplete: false, samplesConsumed: this.samplePosition, totalSamples: this.pendingSignal.length };
  }
  isModulating() { return !!this.pendingSignal; }
  getProgress() { return this.pendingSignal ? this.samplePosition / this.pendingSignal.length : 0; }
  cancel() { this.reset(); }
  reset() { this.pendingSignal = null; this.samplePosition = 0; }
}
module.exports = { ChunkedModulator };
