// Golden-vector harness for HOSTILE input (TEST INFRASTRUCTURE, build container only; round 6, VERDICT r05 #2).
//
// What the REAL reference FSKCore (type-stripped into a temp dir by strip_ts.py, never committed) does with samples a
// microphone path can produce but its own tests never feed it: NaN, +-Inf, |x| >> 1, subnormal floats.  Reference lines
// involved: AGC fsk.ts:52-76 (NaN fails both level tests: the gain holds), pre-filter filters.ts:47-76 (never reset,
// fsk.ts:175-188: one NaN poisons the instance for good; b1 = 0 times Inf is NaN one sample after an Inf), slicer
// fsk.ts:264 (NaN > 0 is false: bit 0), silence test fsk.ts:285 (NaN < threshold is false: never 'eod' again).
//
// usage: node golden_harness_hostile.js <ref_bundle.js> <out_dir>     (make_golden.py drives it)
'use strict';
const fs = require('fs');
const path = require('path');
const H = require('./golden_harness.js');
const { manifest, arrays, saveArray, concat, scaled, str, mkCore, modulate, demodCase } = H;
manifest.generator = 'oracle/refrun/golden_harness_hostile.js';

// the status numbers as doubles (JSON has no NaN / Infinity): frameStarted, globalSampleCounter, receivedBitsLength,
// byteBufferLength, demodulationCalls, syncDetections, silenceThreshold, totalSamplesProcessed, AGC gain
function statusVector(c) {
  const s = c.status;
  return Float64Array.from([s.frameStarted ? 1 : 0, s.globalSampleCounter, s.receivedBitsLength, s.byteBufferLength, s.demodulationCalls,
    s.syncDetections, s.silenceThreshold, s.totalSamplesProcessed, c.agc_gain === null ? 0 : c.agc_gain]);
}
async function hostile(name, cfg, input, chunks, opts) {
  const c = await demodCase(name, cfg, input, chunks, opts);
  c.status_vector = saveArray(name + '.statusv', statusVector(c));
  return c;
}
function withSample(sig, at, bits) {      // one sample replaced by the float with this bit pattern
  const o = new Float32Array(sig);
  new Uint32Array(o.buffer)[at] = bits >>> 0;
  return o;
}

async function main() {
  const BELL = { baudRate: 1200, markFrequency: 1200, spaceFrequency: 2200 };
  const QNAN = 0x7FC00000, NQNAN = 0xFFC00000, SNAN = 0x7FA00001, PINF = 0x7F800000, NINF = 0xFF800000, FMAX = 0x7F7FFFFF;
  for (const [tag, cfg] of [['dflt', {}], ['bell', BELL]]) {
    const a = await modulate(cfg, str('Hello')), b = await modulate(cfg, str('World'));
    const gap = new Float32Array(2000);
    const clean = concat([a, gap, b]);
    await hostile('h_' + tag + '_clean', cfg, clean, 0, {});
    // one bad sample mid-payload of the first frame (after its first byte), then a second, clean frame
    for (const [nm, bits] of [['qnan', QNAN], ['neg_qnan', NQNAN], ['snan', SNAN], ['pinf', PINF], ['ninf', NINF]]) {
      const x = withSample(clean, 1800, bits);
      await hostile('h_' + tag + '_' + nm + '_mid', cfg, x, 0, { trace: (nm === 'qnan' || nm === 'pinf') });
      if (nm === 'qnan' || nm === 'ninf') await hostile('h_' + tag + '_' + nm + '_mid_c128', cfg, x, 128, { inputRef: 'h_' + tag + '_' + nm + '_mid.in' });
    }
    // ... during the idle silence before any frame
    for (const [nm, bits] of [['qnan', QNAN], ['pinf', PINF]]) {
      await hostile('h_' + tag + '_' + nm + '_idle', cfg, withSample(concat([gap, clean]), 1000, bits), 0, {});
    }
    // ... in the gap between the frames, and as the very first / very last sample of a call
    await hostile('h_' + tag + '_qnan_gap', cfg, withSample(clean, a.length + 700, QNAN), 0, {});
    await hostile('h_' + tag + '_qnan_first', cfg, withSample(clean, 0, QNAN), 0, {});
    await hostile('h_' + tag + '_qnan_last_of_call', cfg, withSample(clean, 2047, NQNAN), 2048, {});
    // the largest finite float once: finite all the way in the reference's doubles
    await hostile('h_' + tag + '_fmax_mid', cfg, withSample(clean, 1800, FMAX), 0, {});
    // whole first frame scaled far outside [-1, 1]
    for (const [nm, k] of [['1e10', 1e10], ['1e18', 1e18], ['1e25', 1e25], ['3e38', 3e38]]) {
      await hostile('h_' + tag + '_scale_' + nm, cfg, concat([scaled(a, k), gap, b]), 0, {});
    }
    // ... and down into the subnormal floats (1e-40) and to the last subnormal bit (1e-44 -> a few ulps of 1.4e-45)
    for (const [nm, k] of [['1e-20', 1e-20], ['1e-30', 1e-30], ['1e-40', 1e-40], ['1e-44', 1e-44]]) {
      await hostile('h_' + tag + '_scale_' + nm, cfg, concat([scaled(a, k), gap, b]), 0, { trace: nm === '1e-40' });
    }
  }
  manifest.arrays = arrays;
  fs.writeFileSync(path.join(H.OUT, 'manifest.json'), JSON.stringify(manifest));
  console.log('hostile cases', manifest.cases.length, 'arrays', Object.keys(arrays).length);
}
main().catch((e) => { console.error(e); process.exit(1); });
