'use strict';
// Behavioural KATs of the reference's own vitest suites, replayed against the JS host class
// (napi/fsk-core.js -> N-API -> libfskhip.so).  usage: node fsk_core_test.js cpu|gpu
const assert = require('assert');
const path = require('path');
const M = require(path.join(__dirname, '..', '..', 'napi', 'fsk-core.js'));
const mode = process.argv[2] || 'cpu';

function str(s) { return Uint8Array.from(Buffer.from(s, 'ascii')); }
async function rejects(p, re) {
  let err = null;
  try { await p; } catch (e) { err = e; }
  assert.ok(err && re.test(err.message), 'expected rejection ' + re + ', got ' + (err && err.message));
}

async function cpuTests() {
  assert.strictEqual(M.addon.abiVersion, 8);
  const core = new M.FSKCore();
  assert.strictEqual(core.name, 'FSK');
  assert.strictEqual(core.isReady(), false);
  // tests/modems/fsk-demodulation.node.test.ts:31-36, fsk-modulation 211-216
  await rejects(core.demodulateData(new Float32Array([0.1, 0.2, 0.3])), /not configured/);
  await rejects(core.modulateData(str('x')), /not configured/);
  assert.deepStrictEqual(M.DEFAULT_FSK_CONFIG.preamblePattern, [0x55, 0x55]);
  if (M.addon.deviceCount() === 0) {
    // no GPU: configuring must fail loudly, there is no JavaScript/CPU fallback
    assert.throws(() => core.configure({}), /no CPU fallback/);
    assert.strictEqual(core.isReady(), false);
  }
  console.log('js cpu tests ok');
}

async function gpuTests() {
  for (const precision of [M.PRECISION_F64, M.PRECISION_F32]) {
    const core = new M.FSKCore({ precision });
    const events = [];
    core.on('configured', () => events.push('configured'));
    core.on('eod', () => events.push('eod'));
    core.configure({});
    assert.deepStrictEqual(events, ['configured']);
    // fsk-modulation 75-90: length formula; SURVEY known answers for 'AB'
    const sig = await core.modulateData(str('AB'));
    assert.strictEqual(sig.length, 2480);
    assert.ok(Math.abs(sig[81] - 0.2398044615983963) < 1e-7 && sig[80] === 0);
    // fsk-demodulation 81-106: exact round trip, one sync, one eod
    const buf = Float32Array.from(sig);
    const out = await core.demodulateData(buf);
    assert.deepStrictEqual(Array.from(out), [65, 66]);
    // demodulateData mutates its input when AGC is on (fsk.ts:55); SURVEY known answer at [100]
    assert.ok(Math.abs(buf[100] - (-1.0560816526412964)) < 2e-6, 'AGC write-back ' + buf[100]);
    let st = core.getStatus();
    assert.strictEqual(st.syncDetections, 1);
    assert.strictEqual(st.globalSampleCounter, 25);
    assert.strictEqual(st.receivedBitsLength, 1240);
    assert.strictEqual(st.totalSamplesProcessed, 2480);
    assert.ok(Math.abs(st.silenceThreshold - 0.16749451808631421) < (precision === M.PRECISION_F64 ? 1e-12 : 2e-6));
    assert.strictEqual(events.filter((e) => e === 'eod').length, 1);
    // getSignalQuality(): the reference's zeros (fsk.ts:471-479); the opt-in estimates are an extension (include/fskhip.h)
    assert.deepStrictEqual(core.getSignalQuality(), { snr: 0, ber: 0, eyeOpening: 0, phaseJitter: 0, frequencyOffset: 0 });
    {
      const q = new M.FSKCore({ precision });
      q.configure({});
      q.enableSignalQualityEstimates();
      const msg = await q.modulateData(str('signal quality estimates, please'));
      const padded = new Float32Array(msg.length + 1200);
      padded.set(msg, 100);
      assert.strictEqual(Buffer.from(await q.demodulateData(padded)).toString('ascii'), 'signal quality estimates, please');
      const est = q.getSignalQualityEstimates();
      assert.strictEqual(est.frames, 1);
      assert.strictEqual(est.bytes, 32);
      assert.ok(est.snr > 60 && est.eyeOpening > 0.5 && est.eyeOpening <= 1 && est.signalLevel > 0.1, JSON.stringify(est));
      assert.ok(Math.abs(est.frequencyOffset) < 40, JSON.stringify(est));
      assert.deepStrictEqual(q.getSignalQuality(), { snr: 0, ber: 0, eyeOpening: 0, phaseJitter: 0, frequencyOffset: 0 });
      q.close();
    }
    // configure() on a configured instance (fsk.ts:133-157): resetState() semantics, silence.threshold and the debug
    // counters survive, the AGC is new
    const thrBefore = st.silenceThreshold;
    core.configure({ markFrequency: 1200, spaceFrequency: 2200 });
    st = core.getStatus();
    assert.strictEqual(st.silenceThreshold, thrBefore);
    assert.strictEqual(st.syncDetections, 1);
    assert.strictEqual(st.demodulationCalls, 1);
    assert.strictEqual(st.globalSampleCounter, 0);
    assert.strictEqual(st.frameStarted, false);
    core.configure({});
    // reset keeps ready (fsk.ts:464-469), clears counters
    core.reset();
    assert.strictEqual(core.isReady(), true);
    st = core.getStatus();
    assert.strictEqual(st.syncDetections, 0);
    assert.strictEqual(st.demodulationCalls, 0);
    // fsk-demodulation 363-398: 128-sample chunks
    const hello = await core.modulateData(str('Hello'));
    let got = [];
    for (let off = 0; off < hello.length; off += 128) {
      const chunk = Float32Array.from(hello.subarray(off, Math.min(off + 128, hello.length)));
      got = got.concat(Array.from(await core.demodulateData(chunk)));
    }
    assert.deepStrictEqual(got, Array.from(str('Hello')));
    // fsk-sfd 163-171: empty payload -> no bytes
    core.reset();
    assert.strictEqual((await core.demodulateData(await core.modulateData(new Uint8Array(0)))).length, 0);
    core.close();
  }
  // batch: per-stream tones (BASELINE config #4 shape)
  const S = 5;
  const cfgs = [];
  for (let s = 0; s < S; s++) cfgs.push({ baudRate: 300, markFrequency: 1000 + 10 * s, spaceFrequency: 1200 + 10 * s });
  const batch = new M.FSKBatch(S, cfgs);
  const payloads = [];
  for (let s = 0; s < S; s++) payloads.push(str('s' + s));
  const sigs = batch.modulateData(payloads);
  const N = sigs[0].length;
  const flat = new Float32Array(S * N);
  sigs.forEach((x, s) => flat.set(x, s * N));
  const r = batch.demodulateData(flat, N, N, false);
  for (let s = 0; s < S; s++) assert.deepStrictEqual(Array.from(r.bytes[s]), Array.from(payloads[s]));
  batch.close();
  console.log('js gpu tests ok');
}

(mode === 'gpu' ? gpuTests() : cpuTests()).catch((e) => { console.error(e); process.exit(1); });
