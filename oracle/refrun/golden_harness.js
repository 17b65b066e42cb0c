// Golden-vector harness (TEST INFRASTRUCTURE, build container only).
//
// Runs the REAL reference FSKCore (type-stripped into a temp dir by strip_ts.py, never
// committed) under Node and records inputs / outputs / status / selected intermediates.
// The scenarios restate what the reference's own vitest suites pin (SURVEY.md §4):
//   tests/modems/fsk-demodulation.node.test.ts, fsk-modulation, fsk-sfd, fsk-simplesync,
//   fsk-false-positive, fsk-preamble-robustness, tests/dsp/filters*.node.test.ts
// plus seeded-noise and long multi-frame cases the upstream tests do not have.
//
// usage: node golden_harness.js <ref_bundle.js> <out_dir>
'use strict';
const fs = require('fs');
const path = require('path');
const R = require(path.resolve(process.argv[2]));
const OUT = process.argv[3];
fs.mkdirSync(OUT, { recursive: true });

const manifest = { generator: 'oracle/refrun/golden_harness.js', node: process.version, cases: [] };
const arrays = {}; // name -> {dtype, file}

function saveArray(name, arr) {
  if (arrays[name]) throw new Error('dup array ' + name);
  let dtype;
  if (arr instanceof Float32Array) dtype = 'f4';
  else if (arr instanceof Float64Array) dtype = 'f8';
  else if (arr instanceof Uint8Array) dtype = 'u1';
  else if (arr instanceof Int32Array) dtype = 'i4';
  else throw new Error('bad array type for ' + name);
  const file = name + '.' + dtype + '.bin';
  fs.writeFileSync(path.join(OUT, file), Buffer.from(arr.buffer, arr.byteOffset, arr.byteLength));
  arrays[name] = { dtype, n: arr.length };
  return name;
}

// ---- deterministic PRNG (mulberry32) + Box-Muller -------------------------------------------
function rng32(seed) {
  let a = seed >>> 0;
  return function () {
    a = (a + 0x6D2B79F5) >>> 0;
    let t = a;
    t = Math.imul(t ^ (t >>> 15), t | 1);
    t ^= t + Math.imul(t ^ (t >>> 7), t | 61);
    return ((t ^ (t >>> 14)) >>> 0) / 4294967296;
  };
}
function gaussian(rand) {
  let u = 0;
  while (u === 0) u = rand();
  const v = rand();
  return Math.sqrt(-2 * Math.log(u)) * Math.cos(2 * Math.PI * v);
}
// noise power relative to mean-square of the whole buffer (padding included), the reference
// tests' definition: tests/modems/fsk-demodulation.node.test.ts:1184-1205
function addGaussian(sig, snrDb, seed) {
  const rand = rng32(seed);
  let p = 0;
  for (let i = 0; i < sig.length; i++) p += sig[i] * sig[i];
  p /= sig.length;
  const sigma = Math.sqrt(p / Math.pow(10, snrDb / 10));
  const out = new Float32Array(sig.length);
  for (let i = 0; i < sig.length; i++) out[i] = sig[i] + sigma * gaussian(rand);
  return out;
}
function concat(list) {
  let n = 0;
  for (const a of list) n += a.length;
  const out = new Float32Array(n);
  let o = 0;
  for (const a of list) { out.set(a, o); o += a.length; }
  return out;
}
function scaled(sig, g) {
  const out = new Float32Array(sig.length);
  for (let i = 0; i < sig.length; i++) out[i] = sig[i] * g;
  return out;
}
function str(s) { return new Uint8Array(Buffer.from(s, 'ascii')); }

function mkCore(cfg) {
  const f = new R.FSKCore();
  f.configure(Object.assign({}, R.DEFAULT_FSK_CONFIG, cfg || {}));
  return f;
}
async function modulate(cfg, payload) {
  return mkCore(cfg).modulateData(payload);
}

// Runs `input` through ONE core in the given chunking; records per-call bytes + eod counts.
// opts.trace: also record per-decimated-sample bit/amplitude, post-filter in/out, the
// pre-filter output and the AGC-mutated input.
async function demodCase(name, cfg, input, chunks, opts) {
  opts = opts || {};
  const core = opts.core || mkCore(cfg);
  let eod = 0, err = 0;
  if (!opts.core) { core.on('eod', () => { eod++; }); core.on('error', () => { err++; }); }
  const tr = opts.trace ? { bit: [], amp: [], pin: [], pout: [], pre: [] } : null;
  if (tr) {
    const pdb = core.processDownsampledBit.bind(core);
    core.processDownsampledBit = (b, a) => { tr.bit.push(b); tr.amp.push(a); return pdb(b, a); };
    const pf = core.dsp.postFilter, pfp = pf.process.bind(pf);
    pf.process = (x) => { const y = pfp(x); tr.pin.push(x); tr.pout.push(y); return y; };
    const pre = core.dsp.preFilter, prb = pre.processBuffer.bind(pre);
    pre.processBuffer = (buf) => { const o = prb(buf); for (let i = 0; i < o.length; i++) tr.pre.push(o[i]); return o; };
  }
  const work = new Float32Array(input); // the reference mutates its input (AGC in place)
  const calls = [];
  let off = 0, ci = 0;
  while (off < work.length || (work.length === 0 && ci === 0)) {
    const n = Math.min(Array.isArray(chunks) ? chunks[ci % chunks.length] : (chunks || work.length || 1), work.length - off);
    const e0 = eod;
    const bytes = await core.demodulateData(work.subarray(off, off + n));
    calls.push({ n: n, bytes: Array.from(bytes), eod: eod - e0 });
    off += n; ci++;
    if (work.length === 0) break;
  }
  const c = {
    name, kind: 'demod', config: cfg || {}, input: opts.inputRef || saveArray(name + '.in', input),
    chunk: Array.isArray(chunks) ? chunks : (chunks || 0), calls: compactCalls(calls),
    bytes: [].concat.apply([], calls.map(x => x.bytes)),
    eod_total: calls.reduce((s, x) => s + x.eod, 0), errors: err,
    status: core.getStatus(), agc_gain: core.dsp.agc ? core.dsp.agc.currentGain : null,
    src: opts.src || ''
  };
  if (opts.prefixZeros !== undefined) c.prefix_zeros = opts.prefixZeros;
  if (tr) {
    c.trace = {
      bit: saveArray(name + '.bit', Uint8Array.from(tr.bit)),
      amp: saveArray(name + '.amp', Float64Array.from(tr.amp)),
      post_in: saveArray(name + '.pin', Float64Array.from(tr.pin)),
      post_out: saveArray(name + '.pout', Float64Array.from(tr.pout)),
      pre_out: saveArray(name + '.pre', Float32Array.from(tr.pre)),
      agc_out: saveArray(name + '.agc', work)
    };
  }
  if (!opts.noPush) manifest.cases.push(c);
  return c;
}
// only calls that produced bytes or eod are listed (index, bytes, eod); the rest are implied
function compactCalls(calls) {
  const out = [];
  calls.forEach((c, i) => { if (c.bytes.length || c.eod) out.push({ i: i, bytes: c.bytes, eod: c.eod }); });
  return { count: calls.length, nonempty: out };
}

async function main() {
  const D = R.DEFAULT_FSK_CONFIG;
  const C300 = { baudRate: 300 };                                                  // 1650/1850 @300 (fsk-simplesync)
  const V21BAD = { baudRate: 300, markFrequency: 1270, spaceFrequency: 1070 };     // BASELINE config #1 as written
  const V21OK = { baudRate: 300, markFrequency: 1070, spaceFrequency: 1270 };
  const BELL = { baudRate: 1200, markFrequency: 1200, spaceFrequency: 2200 };
  const F2125 = { markFrequency: 2125, spaceFrequency: 2295 };

  // ---------------- filter design KATs (filters.ts:180-314) ----------------
  const fd = [];
  for (const [fc, sr] of [[1200, 48000], [300, 48000], [1200, 44100], [100, 8000], [3000, 48000]]) {
    fd.push({ fn: 'butterworthLowpass', args: [fc, sr], out: R.FilterDesign.butterworthLowpass(fc, sr) });
    fd.push({ fn: 'butterworthHighpass', args: [fc, sr], out: R.FilterDesign.butterworthHighpass(fc, sr) });
  }
  for (const [fc, bw, sr] of [[1750, 2600, 48000], [1170, 800, 48000], [1700, 3400, 48000], [1750, 1000, 48000], [2210, 2570, 48000], [1750, 2600, 44100]]) {
    fd.push({ fn: 'butterworthBandpass', args: [fc, bw, sr], out: R.FilterDesign.butterworthBandpass(fc, bw, sr) });
  }
  for (const [fc, sr, nt] of [[1000, 48000, 51], [1200, 48000, 50], [300, 8000, 11]]) {
    fd.push({ fn: 'sincLowpass', args: [fc, sr, nt], out: R.FilterDesign.sincLowpass(fc, sr, nt) });
  }
  for (const [fc, sr, nt] of [[1000, 48000, 51], [300, 8000, 11], [1000, 48000, 50]]) {
    fd.push({ fn: 'sincHighpass', args: [fc, sr, nt], out: R.FilterDesign.sincHighpass(fc, sr, nt) });
  }
  for (const [fc, bw, sr, nt] of [[1750, 800, 48000, 51], [1000, 500, 8000, 21], [1750, 800, 48000, 50], [1200, 600, 44100, 8]]) {  // incl. even tap counts
    fd.push({ fn: 'sincBandpass', args: [fc, bw, sr, nt], out: R.FilterDesign.sincBandpass(fc, bw, sr, nt) });
  }
  manifest.filter_design = fd;

  // ---------------- filter run KATs (filters.ts:47-87, 125-151) ----------------
  {
    const rand = rng32(0xF117E4);
    const x = new Float32Array(512);
    for (let i = 0; i < x.length; i++) x[i] = 2 * rand() - 1;
    x[0] = 1; // impulse-ish head
    saveArray('filt.x', x);
    const runs = [];
    const mk = [
      ['iir_lp_1200', () => R.FilterFactory.createIIRLowpass(1200, 48000)],
      ['iir_hp_300', () => R.FilterFactory.createIIRHighpass(300, 48000)],
      ['iir_bp_1750_2600', () => R.FilterFactory.createIIRBandpass(1750, 2600, 48000)],
      ['iir_unnormalised', () => new R.IIRFilter([2, 1, 0.5], [2, -0.5, 0.25])],
      ['iir_order1', () => new R.IIRFilter([0.5, 0.5], [1, -0.2])],
      ['iir_order3', () => new R.IIRFilter([0.1, 0.2, 0.3, 0.4], [1, -0.3, 0.2, -0.1])],
      ['fir_lp_1000_51', () => R.FilterFactory.createFIRLowpass(1000, 48000)],
      ['fir_hp_1000_51', () => R.FilterFactory.createFIRHighpass(1000, 48000)],
      ['fir_bp_1750_800_51', () => R.FilterFactory.createFIRBandpass(1750, 800, 48000)],
      ['fir_taps5', () => new R.FIRFilter([0.1, -0.2, 0.3, 0.25, -0.05])]
    ];
    for (const [nm, f] of mk) {
      const flt = f();
      const y64 = new Float64Array(x.length);
      for (let i = 0; i < x.length; i++) y64[i] = flt.process(x[i]);
      flt.reset();
      const y32 = flt.processBuffer(x);
      const co = flt.getCoefficients();
      runs.push({ name: nm, coeffs: co, x: 'filt.x', y_process: saveArray('filt.' + nm + '.y64', y64), y_buffer: saveArray('filt.' + nm + '.y32', y32) });
    }
    manifest.filter_runs = runs;
  }

  // ---------------- modulation KATs (fsk.ts:377-424; tests/modems/fsk-modulation.node.test.ts) ----------------
  const mods = [];
  async function modCase(name, cfg, payload) {
    const sig = await modulate(cfg, payload);
    mods.push({ name, config: cfg || {}, payload: Array.from(payload), n: sig.length, signal: saveArray('mod.' + name, sig) });
    return sig;
  }
  const sigAB = await modCase('default_AB', {}, str('AB'));
  const sigHello = await modCase('default_Hello', {}, str('Hello'));
  const sigHello300 = await modCase('b300_Hello', C300, str('Hello'));
  await modCase('default_empty', {}, new Uint8Array(0));
  await modCase('default_noframe_empty', { preamblePattern: [], sfdPattern: [] }, new Uint8Array(0));
  await modCase('bell202_HelloWorld', BELL, str('Hello, World!'));
  await modCase('v21ok_H', V21OK, str('H'));
  await modCase('parity_even_AB', { parity: 'even' }, str('AB'));
  await modCase('parity_odd_AB', { parity: 'odd' }, str('AB'));
  await modCase('stop2_AB', { stopBits: 2 }, str('AB'));
  await modCase('sr44100_b300_A', { sampleRate: 44100, baudRate: 300 }, str('A'));
  await modCase('f2125_AB', F2125, str('AB'));
  {
    const all = new Uint8Array(256);
    for (let i = 0; i < 256; i++) all[i] = i;
    await modCase('default_all256', {}, all);
  }
  manifest.modulate = mods;

  // ---------------- demodulation cases ----------------
  // fsk-demodulation.node.test.ts:81-106
  await demodCase('d_default_AB', {}, sigAB, 0, { trace: true, src: 'fsk-demodulation 81-106' });
  // :14-29 empty / very short
  await demodCase('d_empty', {}, new Float32Array(0), 0, { src: 'fsk-demodulation 14-20' });
  await demodCase('d_short100', {}, new Float32Array(100), 0, { src: 'fsk-demodulation 22-29' });
  // :363-398 128-sample chunks
  await demodCase('d_default_Hello_c128', {}, sigHello, 128, { src: 'fsk-demodulation 363-398' });
  await demodCase('d_default_Hello_whole', {}, sigHello, 0, { trace: true });
  // :718-753 chunk sizes
  for (const cs of [32, 64, 256, 1, 7, 1000]) {
    await demodCase('d_default_Hello_c' + cs, {}, sigHello, cs, { inputRef: 'd_default_Hello_c128.in', src: 'fsk-demodulation 718-753' });
  }
  // :400-437 silence then signal
  await demodCase('d_silence2000_AB', {}, concat([new Float32Array(2000), sigAB]), 128, { src: 'fsk-demodulation 400-437' });
  // :668-716 all 128 offsets (input = k zeros ++ default 'Hello'), chunk 128
  {
    const sig = await modulate({}, str('Hi'));
    saveArray('d_offsets.base', sig);
    const offs = [];
    for (let k = 0; k < 128; k++) {
      const c = await demodCase('d_offset_' + k, {}, concat([new Float32Array(k), sig]), 128, { noPush: true, inputRef: 'd_offsets.base' });
      offs.push({ k: k, bytes: c.bytes, eod_total: c.eod_total, status: c.status });
    }
    manifest.offset_sweep = { config: {}, base: 'd_offsets.base', chunk: 128, payload: Array.from(str('Hi')), runs: offs, src: 'fsk-demodulation 668-716' };
  }
  // :854-925 three messages, 500-sample gaps, same instance
  {
    const a = await modulate({}, str('One')), b = await modulate({}, str('Two')), c = await modulate({}, str('Three'));
    const gap = new Float32Array(500);
    await demodCase('d_three_msgs_gap500', {}, concat([a, gap, b, gap, c]), 128, { src: 'fsk-demodulation 854-925' });
  }
  // :1110-1131 single byte patterns
  for (const b of [0x48, 0x55, 0x7E, 0xAA, 0x00, 0xFF, 0x33, 0xF0, 0x0F]) {
    await demodCase('d_byte_' + b.toString(16), {}, await modulate({}, new Uint8Array([b])), 0, { src: 'fsk-demodulation 1110-1131' });
  }
  // :1133-1161 identical bytes, exactly one eod
  await demodCase('d_identical_x3', {}, await modulate({}, new Uint8Array([0x55, 0x55, 0x55])), 0, { src: 'fsk-demodulation 1133-1161' });
  // :493-521 AGC at x0.1 ; :240-259 amplitude variations ; :217-238 DC offset
  await demodCase('d_amp_0p1_c128', {}, scaled(sigHello, 0.1), 128, { src: 'fsk-demodulation 493-521' });
  await demodCase('d_amp_0p01', {}, scaled(sigHello, 0.01), 0, {});
  await demodCase('d_amp_2p0', {}, scaled(sigHello, 2.0), 0, {});
  await demodCase('d_amp_0p5_noagc', { agcEnabled: false }, scaled(sigHello, 0.5), 0, {});
  {
    const dc = new Float32Array(sigHello.length);
    for (let i = 0; i < dc.length; i++) dc[i] = sigHello[i] + 0.1;
    await demodCase('d_dc_offset', {}, dc, 0, { src: 'fsk-demodulation 217-238' });
  }
  // :300-346 baud rates and frequency pairs
  await demodCase('d_b300_Hello', C300, sigHello300, 0, { trace: true, src: 'fsk-simplesync 25-63' });
  await demodCase('d_b300_Hello_c128', C300, sigHello300, 128, { inputRef: 'd_b300_Hello.in' });
  await demodCase('d_f2125_AB', F2125, await modulate(F2125, str('AB')), 0, { src: 'fsk-demodulation 322-346' });
  await demodCase('d_b300_bytes_55557E48', C300, await modulate(C300, new Uint8Array([0x55, 0x55, 0x7E, 0x48])), 0, { src: 'fsk-simplesync 105-116' });
  // BASELINE config #1: V.21 as written decodes nothing; the mark<space twin decodes
  {
    const payload = str('V.21 test');
    const bad = await modulate(V21BAD, payload), ok = await modulate(V21OK, payload);
    const pad = (s) => { const o = new Float32Array(48000); o.set(s.subarray(0, Math.min(s.length, 48000))); return o; };
    await demodCase('d_c1_v21_as_written', V21BAD, pad(bad), 0, { src: 'BASELINE config #1 (mark 1270 / space 1070): reference decodes nothing' });
    await demodCase('d_c1_v21_swapped', V21OK, pad(ok), 0, { src: 'BASELINE config #1 twin (mark 1070 / space 1270)' });
  }
  await demodCase('d_bell202_HelloWorld', BELL, await modulate(BELL, str('Hello, World!')), 0, { trace: true });
  // fsk-sfd: 0x55 / 0x7E payloads, two frames -> two eod, empty payload
  {
    const a = await modulate({}, new Uint8Array([0x55])), b = await modulate({}, new Uint8Array([0x48]));
    await demodCase('d_two_frames', {}, concat([a, b]), 0, { src: 'fsk-sfd 139-159' });
    await demodCase('d_empty_payload', {}, await modulate({}, new Uint8Array(0)), 0, { src: 'fsk-sfd 163-171' });
    // back-to-back frames without the trailing silence (fsk-preamble-robustness 224-262)
    const spb = 40;
    await demodCase('d_back_to_back', {}, concat([a.subarray(0, a.length - 10 * spb), b]), 0, { src: 'fsk-preamble-robustness 224-262' });
  }
  // fsk-false-positive 14-206
  {
    const N = 8192;
    const mkbuf = (f) => { const o = new Float32Array(N); for (let i = 0; i < N; i++) o[i] = f(i); return o; };
    await demodCase('d_fp_zeros', {}, new Float32Array(12000), 0, { src: 'fsk-false-positive' });
    await demodCase('d_fp_dc_pos', {}, mkbuf(() => 0.5), 0, {});
    await demodCase('d_fp_dc_neg', {}, mkbuf(() => -0.5), 0, {});
    await demodCase('d_fp_tone2k', {}, mkbuf((i) => Math.sin(2 * Math.PI * 2000 * i / 48000)), 0, {});
    await demodCase('d_fp_alternating', {}, mkbuf((i) => (i & 1) ? -1 : 1), 0, {});
    const rand = rng32(0xBADF00D);
    await demodCase('d_fp_uniform_noise', {}, mkbuf(() => (rand() - 0.5) * 0.2), 0, { src: 'fsk-simplesync 138-150' });
    await demodCase('d_fp_uniform_noise_b300', C300, mkbuf(() => (rand() - 0.5) * 2.0), 0, {});
    // truncated preamble (fsk-preamble-robustness 65-84): drop the first 75% of the preamble samples
    const cut = Math.floor(2 * 10 * 40 * 0.75);
    await demodCase('d_fp_truncated_preamble', {}, sigHello.subarray(2 * 40 + cut), 0, { src: 'fsk-preamble-robustness 65-84' });
    // mark tone only / space tone only
    await demodCase('d_fp_mark_tone', {}, mkbuf((i) => Math.sin(2 * Math.PI * 1650 * i / 48000)), 0, {});
    await demodCase('d_fp_space_tone', {}, mkbuf((i) => Math.sin(2 * Math.PI * 1850 * i / 48000)), 0, {});
  }
  // framing variants
  await demodCase('d_parity_even', { parity: 'even' }, await modulate({ parity: 'even' }, str('Par')), 0, {});
  await demodCase('d_parity_odd', { parity: 'odd' }, await modulate({ parity: 'odd' }, str('Par')), 0, {});
  await demodCase('d_stop2', { stopBits: 2 }, await modulate({ stopBits: 2 }, str('Stop')), 0, {});
  await demodCase('d_sync_thr_0p7', { syncThreshold: 0.7 }, sigHello, 0, { inputRef: 'd_default_Hello_c128.in' });
  await demodCase('d_sync_thr_0p95', { syncThreshold: 0.95 }, sigHello, 0, { inputRef: 'd_default_Hello_c128.in' });
  await demodCase('d_prefilter_bw_4000', { preFilterBandwidth: 4000 }, sigHello, 0, { inputRef: 'd_default_Hello_c128.in' });
  await demodCase('d_preamble_AA_sfd_D5', { preamblePattern: [0xAA, 0xAA, 0xAA], sfdPattern: [0xD5] },
    await modulate({ preamblePattern: [0xAA, 0xAA, 0xAA], sfdPattern: [0xD5] }, str('eth')), 0, {});
  // fractional ring capacity (SURVEY H6f): 44.1 kHz
  await demodCase('d_sr44100_b1200', { sampleRate: 44100 }, await modulate({ sampleRate: 44100 }, str('44k1')), 0, {});
  {
    const s = await modulate({ sampleRate: 44100 }, str('late'));
    await demodCase('d_sr44100_late_frame', { sampleRate: 44100 }, concat([new Float32Array(4000), s, new Float32Array(500), s]), 0, {});
  }
  // public reset() between messages (fsk-demodulation 281-299)
  {
    const core = mkCore({});
    let eod = 0; core.on('eod', () => { eod++; });
    const a = await core.demodulateData(new Float32Array(sigHello.subarray(0, 1500)));
    core.reset();
    const st1 = core.getStatus();
    const b = await core.demodulateData(new Float32Array(sigAB));
    manifest.reset_case = {
      config: {}, first: 'd_default_Hello_c128.in', first_n: 1500, first_bytes: Array.from(a), status_after_reset: st1,
      second: 'd_default_AB.in', second_bytes: Array.from(b), eod_total: eod, status: core.getStatus(), agc_gain: core.dsp.agc.currentGain
    };
  }
  // seeded gaussian noise (SURVEY §8a noise behaviour; BASELINE config #5 shape)
  {
    const rand = rng32(0x5EED);
    for (const [tag, cfg] of [['bell', BELL], ['dflt', {}], ['v21', V21OK]]) {
      for (const snr of [20, 10, 6]) {
        for (let t = 0; t < 3; t++) {
          const payload = new Uint8Array(16);
          for (let i = 0; i < 16; i++) payload[i] = Math.floor(rand() * 256);
          const clean = await modulate(cfg, payload);
          const noisy = addGaussian(clean, snr, 1000 * snr + t + (tag === 'bell' ? 7 : tag === 'dflt' ? 13 : 29));
          const c = await demodCase('d_noise_' + tag + '_' + snr + 'dB_' + t, cfg, noisy, 0, { trace: (t === 0 && snr === 10 && tag !== 'v21') });
          c.payload = Array.from(payload);
        }
      }
    }
  }
  // long multi-frame buffers: the bench workload shape (lead-in zeros, back-to-back frames, amplitude scale)
  {
    const rand = rng32(0xF5C0DE);
    for (const [tag, cfg, plen, N] of [['bell', BELL, 100, 96000], ['v21', V21OK, 32, 120000], ['dflt', {}, 100, 96000]]) {
      const parts = [new Float32Array(Math.floor(rand() * 400))];
      let n = parts[0].length;
      const payloads = [];
      while (n < N) {
        const p = new Uint8Array(plen);
        for (let i = 0; i < plen; i++) p[i] = Math.floor(rand() * 256);
        const s = await modulate(cfg, p);
        parts.push(s); n += s.length; payloads.push(Array.from(p));
      }
      const buf = scaled(concat(parts).subarray(0, N), 0.1 + 0.9 * rand());
      const c = await demodCase('d_long_' + tag, cfg, buf, 0, {});
      c.payloads = payloads;
      await demodCase('d_long_' + tag + '_c128', cfg, buf, 128, { inputRef: 'd_long_' + tag + '.in' });
    }
  }
  manifest.arrays = arrays;
  fs.writeFileSync(path.join(OUT, 'manifest.json'), JSON.stringify(manifest));
  console.log('cases', manifest.cases.length, 'arrays', Object.keys(arrays).length);
}
// (round 6) golden_harness_hostile.js re-uses the helpers above: loaded as a module this file only exports them
if (require.main === module) main().catch((e) => { console.error(e); process.exit(1); });
else module.exports = { R, OUT, manifest, arrays, saveArray, rng32, gaussian, addGaussian, concat, scaled, str, mkCore, modulate, demodCase };
