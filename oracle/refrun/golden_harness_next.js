// Golden-vector harness for the "next" rows of SURVEY.md 8(f) (TEST INFRASTRUCTURE, build container only).
//
// Runs the REAL reference classes (type-stripped into a temp dir by strip_ts.py, never committed) under Node:
//   CRC16 (src/utils/crc16.ts), XModemPacket (src/transports/xmodem/packet.ts), ChunkedModulator
//   (src/webaudio/chunked-modulator.ts), RingBuffer (src/utils.ts) + FSKCore in 128-sample quanta as
//   FSKProcessor drives them (src/webaudio/processors/fsk-processor.ts:152-167, 256-322).
// FSKProcessor itself needs AudioWorklet globals Node does not have, so the few lines of glue around the real FSKCore /
// RingBuffer / ChunkedModulator are restated here, each block citing what it follows.  XModemTransport runs as is
// (xmodem.ts, with stand-ins for three standard web APIs Node 12 lacks -- see scanBurst).
//
// usage: node golden_harness_next.js <ref_bundle.js> <out_dir>
'use strict';
const fs = require('fs');
const path = require('path');
const R = require(path.resolve(process.argv[2]));
const OUT = process.argv[3];
fs.mkdirSync(OUT, { recursive: true });

const manifest = { generator: 'oracle/refrun/golden_harness_next.js', node: process.version };
const arrays = {};
function saveArray(name, arr) {
  if (arrays[name]) throw new Error('dup array ' + name);
  let dtype;
  if (arr instanceof Float32Array) dtype = 'f4';
  else if (arr instanceof Float64Array) dtype = 'f8';
  else if (arr instanceof Uint8Array) dtype = 'u1';
  else if (arr instanceof Int32Array) dtype = 'i4';
  else throw new Error('bad array type for ' + name);
  fs.writeFileSync(path.join(OUT, name + '.' + dtype + '.bin'), Buffer.from(arr.buffer, arr.byteOffset, arr.byteLength));
  arrays[name] = { dtype, n: arr.length };
  return name;
}
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
function randBytes(rand, n) {
  const p = new Uint8Array(n);
  for (let i = 0; i < n; i++) p[i] = Math.floor(rand() * 256);
  return p;
}
function packRagged(name, list) {
  const off = new Int32Array(list.length + 1);
  let n = 0;
  list.forEach((a, i) => { off[i] = n; n += a.length; });
  off[list.length] = n;
  const data = new Uint8Array(n);
  list.forEach((a, i) => data.set(a, off[i]));
  return { data: saveArray(name + '.data', data), off: saveArray(name + '.off', off) };
}
function mkCore(cfg) {
  const f = new R.FSKCore();
  f.configure(Object.assign({}, R.DEFAULT_FSK_CONFIG, cfg || {}));
  return f;
}

// ---- the receive grammar of XModemTransport over a recorded byte burst ---------------------------
// The REAL class (xmodem.ts, type-stripped by strip_ts.py): receiveData() is run against a scripted data channel that
// delivers the burst and then nothing, like the reference's own tests drive it (tests/transports/xmodem/xmodem.node.test.ts).
// Node 12 lacks three standard web APIs the class uses (AbortController xmodem.ts:55,73; AbortSignal.timeout / .any
// xmodem.ts:536-542): minimal stand-ins with the standard semantics, nothing of the reference's.
class MiniSignal {
  constructor() { this.aborted = false; this.reason = undefined; this._l = []; }
  addEventListener(t, f) { if (t === 'abort') this._l.push(f); }
  removeEventListener(t, f) { this._l = this._l.filter(x => x !== f); }
  _fire(reason) { if (this.aborted) return; this.aborted = true; this.reason = reason; this._l.slice().forEach(f => f()); }
}
global.AbortController = class { constructor() { this.signal = new MiniSignal(); } abort(r) { this.signal._fire(r || new Error('This operation was aborted')); } };
global.AbortSignal = {
  timeout(ms) { const s = new MiniSignal(); const t = setTimeout(() => s._fire(new Error('The operation was aborted due to timeout')), ms); return s; },
  any(list) { const s = new MiniSignal(); for (const x of list) { if (x.aborted) { s._fire(x.reason); break; } x.addEventListener('abort', () => s._fire(x.reason)); } return s; },
};
class ScriptChannel {        // IDataChannel (core.ts:45-86): modulate records what the transport sends, demodulate hands out the script
  constructor(chunks) { this.q = chunks.slice(); this.sent = []; }
  async modulate(data) { this.sent.push(Array.from(data)); }
  async demodulate(options) {
    if (this.q.length) return this.q.shift();
    return new Promise((resolve, reject) => {   // nothing more arrives: only the caller's timeout signal ends the wait
      const sig = options && options.signal;
      if (!sig) return;
      if (sig.aborted) { reject(new Error('Demodulation aborted')); return; }
      sig.addEventListener('abort', () => reject(new Error('Demodulation aborted')));
    });
  }
  reset() {}
}
// One burst through XModemTransport.receiveData() (xmodem.ts:184-214 -> receiveAllPackets 232-264 -> receiveAndProcessPacket
// 266-320) with maxRetries 0, so that the first error (or the timeout once the bytes run out) ends the call like one scan.
// Instrumentation only: the starting sequence number is planted after initializeReceive() (a receiver that is already
// `expected - 1` packets into a transfer), and receiveAndProcessPacket is wrapped to know whether a timeout hit mid-packet.
async function scanBurst(bytes, expected, split) {
  const log = console.log, warn = console.warn;
  console.log = () => {}; console.warn = () => {};
  try {
    const chunks = [];
    if (split && bytes.length) { for (let i = 0; i < bytes.length; i += split) chunks.push(bytes.slice(i, i + split)); } else if (bytes.length) chunks.push(bytes);
    const ch = new ScriptChannel(chunks);
    const t = new R.XModemTransport(ch);
    t.configure({ timeoutMs: 10, maxRetries: 0 });
    const init = t.initializeReceive;
    t.initializeReceive = function () { init.call(this); this.receive.expectedSequence = expected; };
    let inPacket = false, header = null;
    const rap = t.receiveAndProcessPacket;
    t.receiveAndProcessPacket = async function (sig) { inPacket = true; const r = await rap.call(this, sig); inPacket = false; return r; };
    const wfb = t.waitForBytes;
    t.waitForBytes = async function (count, opt) { const r = await wfb.call(this, count, opt); if (inPacket && count === 3) header = Array.from(r); return r; };
    const errors = [];
    t.on('error', ev => errors.push(ev.data));
    let result = null, err = null;
    try { result = await t.receiveData(); } catch (e) { err = e.message; }
    const st = t.getStatistics();
    let status;
    if (result) status = 'eot';
    else if (/Invalid sequence number/.test(err)) status = 'invalid_sequence';
    else if (/Invalid CRC/.test(err)) status = 'invalid_crc';
    else if (/Unexpected sequence number/.test(err)) status = 'unexpected_sequence';
    else if (/aborted/i.test(err)) status = inPacket ? 'truncated' : 'need_more';
    else throw new Error('unclassified receive error: ' + err);
    let data = result;
    if (!data) { let n = 0; t.receive.data.forEach(p => { n += p.length; }); data = new Uint8Array(n); let o = 0; t.receive.data.forEach(p => { data.set(p, o); o += p.length; }); }
    const e0 = errors[0] || {};
    const errSeq = status === 'truncated' ? (header ? header[0] : -1) : (e0.seq !== undefined ? e0.seq : (e0.received !== undefined ? e0.received : -1));
    const errLen = (status !== 'eot' && status !== 'need_more' && header) ? header[2] : -1;
    const acks = ch.sent.filter(m => m.length === 1 && m[0] === R.ControlType.ACK).length;
    const naks = ch.sent.filter(m => m.length === 1 && m[0] === R.ControlType.NAK).length;
    return { status, expected_after: t.receive.expectedSequence, packets: st.packetsReceived, dropped: st.packetsDropped,
      consumed: bytes.length - t.receive.buffer.length - ch.q.reduce((a, c) => a + c.length, 0), err_seq: errSeq, err_len: errLen,
      crc_rx: e0.crc !== undefined ? e0.crc : -1, crc_calc: e0.calculatedCrc !== undefined ? e0.calculatedCrc : -1, acks, naks, error: err, data };
  } finally { console.log = log; console.warn = warn; }
}

async function main() {
  // ---------------- CRC-16 KATs (crc16.ts:21-38; tests/utils/crc16.node.test.ts) ----------------
  {
    const rand = rng32(0xC4C16);
    const list = [new Uint8Array(0), Uint8Array.from([0x41]), Uint8Array.from([0x31, 0x32, 0x33, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39]),
      Uint8Array.from([0]), Uint8Array.from([0xFF]), Uint8Array.from([0xAA, 0xAA])];
    const ramp = new Uint8Array(256);
    for (let i = 0; i < 256; i++) ramp[i] = i;
    list.push(ramp);
    for (const n of [1, 2, 3, 7, 16, 31, 64, 127, 128, 129, 200, 255, 256, 300, 1000]) list.push(randBytes(rand, n));
    const vals = Int32Array.from(list.map(a => R.CRC16.calculate(a)));
    const packed = packRagged('crc', list);
    manifest.crc = { data: packed.data, off: packed.off, values: saveArray('crc.values', vals),
      verify_true: R.CRC16.verify(list[2], 0x29B1), verify_false: R.CRC16.verify(list[2], 0x29B0) };
  }
  // ---------------- XModem packets (packet.ts:21-53) ----------------
  const pktList = [];
  {
    const rand = rng32(0x9AC4E7);
    const specs = [[1, 0], [1, 1], [2, 3], [255, 128], [17, 255], [128, 64], [254, 5], [3, 100]];
    for (let i = 0; i < 16; i++) specs.push([1 + Math.floor(rand() * 255), Math.floor(rand() * 256)]);
    const payloads = [], wires = [], meta = [];
    for (const [seq, len] of specs) {
      const payload = randBytes(rand, len);
      const p = R.XModemPacket.createData(seq, payload);
      const wire = R.XModemPacket.serialize(p);
      payloads.push(payload); wires.push(wire);
      meta.push({ seq, len, inv: p.invSequence, crc: p.checksum, verify: R.XModemPacket.verify(p) });
      pktList.push({ seq, payload, wire });
    }
    const errors = [];
    for (const [seq, len] of [[0, 1], [256, 1], [-1, 0], [1, 256]]) {
      try { R.XModemPacket.createData(seq, new Uint8Array(len)); errors.push({ seq, len, error: null }); } catch (e) { errors.push({ seq, len, error: e.message }); }
    }
    const pp = packRagged('pkt.payload', payloads), pw = packRagged('pkt.wire', wires);
    manifest.packets = { payload: pp, wire: pw, meta, errors,
      control: { SOH: Array.from(R.XModemPacket.serializeControl(R.ControlType.SOH)), ACK: Array.from(R.XModemPacket.serializeControl(R.ControlType.ACK)),
        NAK: Array.from(R.XModemPacket.serializeControl(R.ControlType.NAK)), EOT: Array.from(R.XModemPacket.serializeControl(R.ControlType.EOT)) },
      constants: R.PacketConstants };
  }
  // ---------------- burst scans (xmodem.ts:233-320 grammar over recorded bytes) ----------------
  {
    const rand = rng32(0x5CA17);
    const mk = (seq, len) => R.XModemPacket.serialize(R.XModemPacket.createData(seq, randBytes(rand, len)));
    const cat = (...parts) => { let n = 0; parts.forEach(p => { n += p.length; }); const o = new Uint8Array(n); let k = 0; parts.forEach(p => { o.set(p, k); k += p.length; }); return o; };
    const flip = (a, i, m) => { const o = Uint8Array.from(a); o[i] ^= m; return o; };
    const bursts = [];
    const add = (name, bytes, expected) => bursts.push({ name, bytes, expected });
    add('empty', new Uint8Array(0), 1);
    add('one_ok', mk(1, 16), 1);
    add('one_ok_len0', mk(1, 0), 1);
    add('one_ok_len255', mk(7, 255), 7);
    add('garbage_then_ok', cat(Uint8Array.from([0x55, 0x00, 0xFF, 0x06, 0x15]), mk(1, 32)), 1);
    add('three_then_eot', cat(mk(1, 128), mk(2, 128), mk(3, 40), Uint8Array.from([0x04])), 1);
    add('eot_only', Uint8Array.from([0x04]), 5);
    add('eot_after_garbage', Uint8Array.from([0x33, 0x04, 0x01]), 1);
    add('no_soh', Uint8Array.from([0x02, 0x03, 0x05, 0x06, 0x15, 0x7E]), 1);
    add('bad_crc_payload', flip(mk(1, 20), 9, 0x10), 1);
    add('bad_crc_field_hi', (() => { const w = mk(1, 20); return flip(w, w.length - 2, 0x01); })(), 1);
    add('bad_crc_field_lo', (() => { const w = mk(1, 20); return flip(w, w.length - 1, 0x80); })(), 1);
    add('bad_nseq', flip(mk(1, 20), 2, 0x04), 1);
    add('bad_seq_both_consistent', mk(9, 20), 1);                 // unexpected sequence
    add('duplicate_then_next', cat(mk(4, 10), mk(5, 12)), 5);     // 4 is previous of 5
    add('duplicate_wrap', cat(mk(255, 10), mk(1, 3)), 1);         // previous of 1 is 255
    add('wrap_255_to_1', cat(mk(254, 4), mk(255, 4), mk(1, 4), mk(2, 4)), 254);
    add('truncated_header', mk(1, 20).slice(0, 3), 1);
    add('truncated_payload', mk(1, 20).slice(0, 15), 1);
    add('truncated_crc', (() => { const w = mk(1, 20); return w.slice(0, w.length - 1); })(), 1);
    add('ok_then_truncated', cat(mk(1, 8), mk(2, 30).slice(0, 12)), 1);
    add('ok_then_bad_crc', cat(mk(1, 8), flip(mk(2, 30), 10, 0x40)), 1);
    add('len_field_corrupt_longer', flip(mk(1, 20), 3, 0x40), 1); // len 20 -> 84: runs out of bytes
    add('len_field_corrupt_shorter', flip(mk(1, 20), 3, 0x04), 1); // len 20 -> 16: CRC taken from payload bytes
    add('soh_inside_garbage', cat(Uint8Array.from([0x01, 0x10]), mk(1, 6)), 1); // a stray SOH eats the header
    add('payload_contains_soh_eot', R.XModemPacket.serialize(R.XModemPacket.createData(1, Uint8Array.from([1, 4, 1, 4, 0x15, 6]))), 1);
    for (let i = 0; i < 24; i++) {
      // random mixtures: a run of packets from a random starting sequence, sometimes corrupted
      let seq = 1 + Math.floor(rand() * 255);
      const start = seq;
      const parts = [];
      const n = 1 + Math.floor(rand() * 5);
      for (let k = 0; k < n; k++) {
        if (rand() < 0.3) parts.push(randBytes(rand, Math.floor(rand() * 4)).map(b => (b === 1 || b === 4) ? 0x20 : b));
        let w = mk(seq, Math.floor(rand() * 140));
        if (rand() < 0.15) w = flip(w, Math.floor(rand() * w.length), 1 << Math.floor(rand() * 8));
        parts.push(w);
        seq = (seq % 255) + 1;
      }
      if (rand() < 0.5) parts.push(Uint8Array.from([0x04]));
      add('rand_' + i, cat(...parts), start);
    }
    const outs = [];
    for (const b of bursts) outs.push(await scanBurst(b.bytes, b.expected, 0));
    // the same bursts delivered 7 bytes at a time must scan identically (waitForBytes accumulates across demodulate() calls)
    for (let i = 0; i < bursts.length; i++) {
      const o7 = await scanBurst(bursts[i].bytes, bursts[i].expected, 7);
      for (const k of ['status', 'expected_after', 'packets', 'dropped', 'consumed', 'err_seq', 'crc_rx', 'crc_calc', 'acks'])
        if (o7[k] !== outs[i][k]) throw new Error('burst ' + bursts[i].name + ': ' + k + ' depends on the chunking: ' + o7[k] + ' vs ' + outs[i][k]);
    }
    const pb = packRagged('scan.bytes', bursts.map(b => b.bytes));
    const pd = packRagged('scan.out', outs.map(o => o.data));
    manifest.scans = { bytes: pb, data: pd, cases: bursts.map((b, i) => {
      const o = outs[i];
      return { name: b.name, expected: b.expected, status: o.status, expected_after: o.expected_after, packets: o.packets,
        dropped: o.dropped, consumed: o.consumed, err_seq: o.err_seq, err_len: o.err_len, crc_rx: o.crc_rx, crc_calc: o.crc_calc,
        acks: o.acks, naks: o.naks, error: o.error };
    }) };
  }
  // ---------------- ChunkedModulator (chunked-modulator.ts; tests/webaudio/chunked-modulator.node.test.ts) ----------------
  {
    const cases = [];
    const BELL = { baudRate: 1200, markFrequency: 1200, spaceFrequency: 2200 };
    for (const [name, cfg, payload, chunk] of [['ab_128', {}, [0x41, 0x42], 128], ['h_128', {}, [0x48], 128], ['abcd_128', {}, [0x41, 0x42, 0x43, 0x44], 128],
      ['u_1', {}, [0x55], 1], ['u_32', {}, [0x55], 32], ['u_64', {}, [0x55], 64], ['u_256', {}, [0x55], 256], ['u_100', {}, [0x55], 100],
      ['bell_hello_128', BELL, [72, 101, 108, 108, 111], 128], ['ab_exact', {}, [0x41, 0x42], 2480], ['ab_over', {}, [0x41, 0x42], 5000], ['ab_half', {}, [0x41, 0x42], 1240]]) {
      const core = mkCore(cfg);
      const cm = new R.ChunkedModulator(core);
      const pre = { modulating: cm.isModulating(), progress: cm.getProgress(), next: cm.getNextSamples(128) };
      await cm.startModulation(Uint8Array.from(payload));
      const direct = await core.modulateData(Uint8Array.from(payload));
      const steps = [];
      const got = [];
      let r;
      const started = cm.isModulating();
      while ((r = cm.getNextSamples(chunk)) !== null) {
        steps.push([r.signal.length, r.isComplete ? 1 : 0, r.samplesConsumed, r.totalSamples, cm.getProgress(), cm.isModulating() ? 1 : 0]);
        for (let i = 0; i < r.signal.length; i++) got.push(r.signal[i]);
        if (steps.length > 100000) throw new Error('runaway');
      }
      let same = got.length === direct.length;
      for (let i = 0; same && i < direct.length; i++) same = got[i] === direct[i];
      // a u_1 trace is 1 880 rows of the same shape: keep head and tail only
      const keep = steps.length > 64 ? steps.slice(0, 8).concat(steps.slice(-8)) : steps;
      cases.push({ name, config: cfg, payload, chunk, pre, started, n_steps: steps.length, steps: keep, steps_truncated: steps.length > 64,
        total: direct.length, identical_to_direct: same, after: { modulating: cm.isModulating(), progress: cm.getProgress(), next: cm.getNextSamples(chunk) } });
    }
    // empty data and cancel
    {
      const core = mkCore({});
      const cm = new R.ChunkedModulator(core);
      await cm.startModulation(new Uint8Array(0));
      const e = { modulating: cm.isModulating(), next: cm.getNextSamples(128), progress: cm.getProgress() };
      await cm.startModulation(Uint8Array.from([1, 2, 3]));
      cm.getNextSamples(128);
      const mid = { modulating: cm.isModulating(), progress: cm.getProgress() };
      cm.cancel();
      const c = { modulating: cm.isModulating(), next: cm.getNextSamples(128), progress: cm.getProgress() };
      // restart while a signal is pending replaces it; empty data while pending cancels it
      await cm.startModulation(Uint8Array.from([1, 2, 3]));
      cm.getNextSamples(300);
      await cm.startModulation(Uint8Array.from([9]));
      const r1 = cm.getNextSamples(128);
      const restart = { consumed: r1.samplesConsumed, total: r1.totalSamples };
      await cm.startModulation(new Uint8Array(0));
      const e2 = { modulating: cm.isModulating(), next: cm.getNextSamples(128) };
      manifest.chunked_misc = { empty: e, mid, cancel: c, restart, empty_while_pending: e2 };
    }
    manifest.chunked = cases;
  }
  // ---------------- FSKProcessor quantum loop (fsk-processor.ts:152-167 process, 256-276 modulateTo, 294-322
  // processDemodulation, 117-138 demodulate(), 228-235 'modulate' clears the RX ring on completion) ----------------
  {
    const BELL = { baudRate: 1200, markFrequency: 1200, spaceFrequency: 2200 };
    const rand = rng32(0xF1F0);
    const runs = [];
    // rx: `frames` frames of `plen` bytes back to back after `lead` zeros; drains at the listed quanta
    for (const [name, cfg, frames, plen, lead, drains, ringCap] of [
      ['rx_bell_3x100_drain', BELL, 3, 100, 320, [400, 700, 100000], 1024],
      ['rx_bell_13x100_overflow', BELL, 13, 100, 56, [100000], 1024],
      ['rx_dflt_4x40_smallring', {}, 4, 40, 0, [250, 100000], 64],
      ['rx_v21_2x16', { baudRate: 300, markFrequency: 1070, spaceFrequency: 1270 }, 2, 16, 1000, [900, 100000], 1024]]) {
      const core = mkCore(cfg);
      const ring = new R.RingBuffer(Uint8Array, ringCap);           // fsk-processor.ts:84
      const parts = [new Float32Array(lead)];
      const payloads = [];
      for (let f = 0; f < frames; f++) {
        const p = randBytes(rand, plen);
        payloads.push(Array.from(p));
        parts.push(await mkCore(cfg).modulateData(p));
      }
      let n = 0;
      parts.forEach(p => { n += p.length; });
      const nq = Math.ceil(n / 128) + 4;
      const buf = new Float32Array(nq * 128);
      let o = 0;
      parts.forEach(p => { buf.set(p, o); o += p.length; });
      const drained = [];
      const lens = [];
      for (let q = 0; q < nq; q++) {
        const bytes = await core.demodulateData(buf.subarray(q * 128, q * 128 + 128));   // 294-322
        for (const b of bytes) ring.put(b);
        if (drains.indexOf(q) >= 0) {
          const m = ring.length;                                                         // 117-138
          const out = [];
          for (let i = 0; i < m; i++) out.push(ring.remove());
          drained.push({ quantum: q, bytes: out });
        }
        if (q % 64 === 0) lens.push([q, ring.length]);
      }
      const m = ring.length;
      const out = [];
      for (let i = 0; i < m; i++) out.push(ring.remove());
      drained.push({ quantum: nq, bytes: out });
      runs.push({ name, kind: 'rx', config: cfg, frames, plen, lead, payloads, quanta: nq, ring_capacity: ringCap, drains: drained, length_probe: lens });
    }
    // tx: a modulation started before quantum q0; output quanta are zeros + slices; completion quantum recorded
    for (const [name, cfg, payload, q0, nq] of [['tx_ab', {}, [0x41, 0x42], 3, 30], ['tx_bell_hello', BELL, [72, 101, 108, 108, 111], 0, 40]]) {
      const core = mkCore(cfg);
      let pending = null;
      const outBuf = new Float32Array(nq * 128);
      let completeAt = -1, generated = 0;
      for (let q = 0; q < nq; q++) {
        if (q === q0) { pending = new R.ChunkedModulator(core); await pending.startModulation(Uint8Array.from(payload)); }
        const outq = outBuf.subarray(q * 128, q * 128 + 128);
        outq.fill(0);                                                                     // 256-276
        if (pending) {
          const r = pending.getNextSamples(outq.length);
          if (r) {
            generated += r.signal.length;
            outq.set(r.signal);
            if (r.isComplete) { pending = null; completeAt = q; }
          }
        }
      }
      const direct = await mkCore(cfg).modulateData(Uint8Array.from(payload));
      let same = true;
      for (let i = 0; i < outBuf.length; i++) {
        const k = i - q0 * 128;
        const want = (k >= 0 && k < direct.length) ? direct[k] : 0;
        if (outBuf[i] !== want) { same = false; break; }
      }
      runs.push({ name, kind: 'tx', config: cfg, payload, start_quantum: q0, quanta: nq, complete_at: completeAt, generated, total: direct.length, output_is_shifted_direct: same });
    }
    manifest.processor = runs;
  }
  // ---------------- Math.sin of the engine the reference runs on (V8's fdlibm port), for oracle/v8_sin.h and the
  // device modulator's fsk_fdlibm.h: uniform samples of the phase range modulateData reaches, plus the corners of
  // fdlibm's argument reduction (multiples of pi/2, the 2^-27 and pi/4 thresholds, the end of the medium range)
  {
    const rand = rng32(0x51115);
    const xs = [0, 1e-9, 7.450580596923828e-9, 0.3, 0.78125, Math.PI / 4, 0.7853981633974484, 2.356194490192345, Math.PI / 2, Math.PI,
      1.5707963267948966, 1.5707963705062866, 3 * Math.PI / 2, 2 * Math.PI, 100 * Math.PI, 823549.6, 823549.66, 1e5, 12345.678];
    for (let n = 1; n <= 40; n++) { xs.push(n * Math.PI / 2); xs.push(n * 1.5707963267948966 * (1 + 1e-15)); xs.push(n * 1.5707963267948966 * (1 - 1e-15)); }
    for (let i = 0; i < 3000; i++) xs.push(rand() * 20000);
    for (let i = 0; i < 1500; i++) xs.push(rand() * 823549);
    for (let i = 0; i < 500; i++) xs.push(-rand() * 1000);
    const x = Float64Array.from(xs);
    const y = new Float64Array(x.length);
    for (let i = 0; i < x.length; i++) y[i] = Math.sin(x[i]);
    manifest.sin = { x: saveArray('sin.x', x), y: saveArray('sin.y', y) };
  }
  manifest.arrays = arrays;
  fs.writeFileSync(path.join(OUT, 'manifest.json'), JSON.stringify(manifest));
  console.log('next-row goldens: arrays', Object.keys(arrays).length);
}
main().catch((e) => { console.error(e); process.exit(1); });
