'use strict';
// The SURVEY 8(f) rows through the JS host classes (napi/*.js -> N-API -> libfskhip.so): the reference's own KATs
// (tests/utils/crc16.node.test.ts, tests/webaudio/chunked-modulator.node.test.ts, tests/dsp/filters.node.test.ts) and
// the golden manifest captured from the real reference classes.  usage: node next_rows_test.js cpu|gpu
const assert = require('assert');
const fs = require('fs');
const path = require('path');
const N = path.join(__dirname, '..', '..', 'napi');
const M = require(path.join(N, 'fsk-core.js'));
const X = require(path.join(N, 'xmodem.js'));
const { ChunkedModulator } = require(path.join(N, 'chunked-modulator.js'));
const { FSKProcessorBatch } = require(path.join(N, 'fsk-processor.js'));
const F = require(path.join(N, 'filters.js'));
const golden = JSON.parse(fs.readFileSync(path.join(__dirname, '..', 'golden', 'manifest_next.json'), 'utf8'));
const mode = process.argv[2] || 'cpu';

async function cpuTests() {
  for (const fn of ['crc16', 'xmodemSerialize', 'xmodemScan', 'processorCreate', 'processorProcess', 'firCreate', 'sincLowpass']) {
    assert.strictEqual(typeof M.addon[fn], 'function', fn);
  }
  // createData's checks come before any device work, with the reference's texts (packet.ts:22-27)
  for (const e of golden.packets.errors) {
    assert.throws(() => X.XModemPacket.createData(e.seq, new Uint8Array(e.len)), (err) => err.message === e.error);
  }
  assert.deepStrictEqual(Object.assign({}, X.ControlType), { SOH: 1, ACK: 6, NAK: 21, EOT: 4 });
  assert.deepStrictEqual(Object.assign({}, X.PacketConstants), golden.packets.constants);
  // designs are host arithmetic: sincLowpass KAT from the reference (symmetric, odd length bumped)
  const lp = F.FilterDesign.sincLowpass(1000, 48000, 50);
  assert.strictEqual(lp.length, 51);
  for (let i = 0; i < 25; i++) assert.ok(Math.abs(lp[i] - lp[50 - i]) < 1e-15);
  if (M.addon.deviceCount() === 0) {
    assert.throws(() => X.CRC16.calculate(Uint8Array.of(1, 2, 3)), /no CPU fallback/);
    assert.throws(() => new F.FIRFilter([0.5, 0.5]), /no CPU fallback/);
    assert.throws(() => new F.IIRFilter([], [1]), /Feedforward coefficients \(b\) cannot be empty/);        // filters.ts:19-21
    assert.throws(() => new F.IIRFilter([1], []), /Feedback coefficients \(a\) cannot be empty/);
    assert.throws(() => new F.IIRFilter([1], [0, 1]), /First feedback coefficient \(a\[0\]\) cannot be zero/);
    assert.throws(() => new F.IIRFilter([1, 2], [1, 0.5]), /no CPU fallback/);
    assert.strictEqual(F.FilterDesign.butterworthBandpass(1750, 2600, 48000).b[0], 0.1430310558532532);   // (host arithmetic: no GPU needed)
  }
  console.log('js next cpu tests ok');
}

async function gpuTests() {
  // ---- CRC16 (tests/utils/crc16.node.test.ts:12-60) ----
  const s = (t) => Uint8Array.from(Buffer.from(t, 'ascii'));
  assert.strictEqual(X.CRC16.calculate(new Uint8Array(0)), 0xFFFF);
  assert.strictEqual(X.CRC16.calculate(s('A')), 0xB915);
  assert.strictEqual(X.CRC16.calculate(s('123456789')), 0x29B1);
  assert.strictEqual(X.CRC16.calculate(Uint8Array.of(0)), 0xE1F0);
  assert.strictEqual(X.CRC16.calculate(Uint8Array.of(0xFF)), 0xFF00);
  assert.strictEqual(X.CRC16.calculate(Uint8Array.of(0xAA, 0xAA)), 0xFB1A);
  assert.strictEqual(X.CRC16.calculate(Uint8Array.from({ length: 256 }, (_, i) => i)), 0x3FBD);
  assert.ok(X.CRC16.verify(s('123456789'), 0x29B1) && !X.CRC16.verify(s('123456789'), 0x29B0));
  // ---- XModemPacket ----
  const pk = X.XModemPacket.createData(3, Uint8Array.of(1, 2, 3));
  assert.deepStrictEqual(Array.from(X.XModemPacket.serialize(pk)), [1, 3, 252, 3, 1, 2, 3, 173, 173]);
  assert.ok(X.XModemPacket.verify(pk));
  golden.packets.meta.slice(0, 8).forEach((m) => {
    const p = X.XModemPacket.createData(m.seq, new Uint8Array(m.len));
    assert.strictEqual(p.invSequence, m.inv);
  });
  // ---- scan: three packets + EOT, a corrupted one, a duplicate ----
  const wires = X.serializeBatch([1, 2, 3], [s('hello'), s('world!'), new Uint8Array(0)]);
  const cat = (...a) => Uint8Array.from([].concat(...a.map((x) => Array.from(x))));
  const bad = Uint8Array.from(wires[0]); bad[5] ^= 0x10;
  const res = X.scanBursts([cat(Uint8Array.of(0x55), wires[0], wires[1], wires[2], Uint8Array.of(4)), bad, cat(wires[0], wires[1])], [1, 1, 2]);
  assert.strictEqual(res[0].statusName, 'eot'); assert.strictEqual(res[0].packets, 3); assert.strictEqual(res[0].expectedAfter, 4);
  assert.strictEqual(Buffer.from(res[0].data).toString('ascii'), 'helloworld!');
  assert.strictEqual(res[1].error, 'Invalid CRC'); assert.strictEqual(res[1].dropped, 1);
  assert.strictEqual(res[2].packets, 1); assert.strictEqual(res[2].dropped, 1);   // seq 1 is the previous of 2: duplicate
  assert.strictEqual(Buffer.from(res[2].data).toString('ascii'), 'world!');
  // ---- ChunkedModulator against the reference's recorded steps ----
  for (const c of golden.chunked) {
    const core = new M.FSKCore();
    core.configure(c.config);
    const cm = new ChunkedModulator(core);
    assert.strictEqual(cm.isModulating(), false); assert.strictEqual(cm.getNextSamples(128), null);
    await cm.startModulation(Uint8Array.from(c.payload));
    const direct = await core.modulateData(Uint8Array.from(c.payload));
    assert.strictEqual(direct.length, c.total);
    const steps = [];
    let r, off = 0;
    while ((r = cm.getNextSamples(c.chunk)) !== null) {
      steps.push([r.signal.length, r.isComplete ? 1 : 0, r.samplesConsumed, r.totalSamples, cm.getProgress(), cm.isModulating() ? 1 : 0]);
      for (let i = 0; i < r.signal.length; i++) assert.strictEqual(r.signal[i], direct[off + i]);
      off += r.signal.length;
    }
    assert.strictEqual(steps.length, c.n_steps);
    assert.deepStrictEqual(c.steps_truncated ? steps.slice(0, 8).concat(steps.slice(-8)) : steps, c.steps, c.name);
    core.close();
  }
  // ---- FSKProcessorBatch: loopback -- stream 0 modulates a packet, its output is fed back as everyone's input ----
  {
    const S = 3;
    const cfg = { baudRate: 1200, markFrequency: 1200, spaceFrequency: 2200 };
    const batch = new M.FSKBatch(S, cfg, { precision: M.PRECISION_F64 });
    const proc = new FSKProcessorBatch(batch, { clearRxOnTxComplete: false });
    const wire = X.serializeBatch([1], [s('loopback over the GPU')])[0];
    proc.modulate([wire, new Uint8Array(0), new Uint8Array(0)], [true, false, false]);
    assert.throws(() => proc.modulate([wire, wire, wire], [true, false, false]), /Modulation already in progress/);
    let st = proc.txState();
    assert.deepStrictEqual(Array.from(st.pending), [1, 0, 0]);
    const total = st.total[0];
    let input = new Float32Array(S * 128), quanta = 0;
    while (quanta < 400) {
      const out = proc.process(input, 128, 128);
      const next = new Float32Array(S * 128);
      for (let k = 0; k < S; k++) next.set(out.subarray(0, 128), k * 128);   // everyone hears stream 0
      input = next; quanta++;
      if (!proc.txState().pending[0] && quanta * 128 > total + 2000) break;
    }
    assert.strictEqual(proc.txState().completed[0], 1);
    assert.deepStrictEqual(Array.from(proc.rxLengths()), [wire.length, wire.length, wire.length]);
    const got = proc.demodulate();
    const scans = X.scanBursts(got, [1, 1, 1]);
    for (const r of scans) { assert.strictEqual(r.packets, 1); assert.strictEqual(Buffer.from(r.data).toString('ascii'), 'loopback over the GPU'); }
    assert.strictEqual(proc.status(1).demodulatedBufferLength, 0);
    proc.reset();
    proc.close(); batch.close();
  }
  // ---- FSKBatch.demodulateDataAsync: same result as the synchronous call, event loop free meanwhile ----
  {
    const S = 16;
    const cfg = { baudRate: 1200, markFrequency: 1200, spaceFrequency: 2200 };
    const a = new M.FSKBatch(S, cfg), b = new M.FSKBatch(S, cfg);
    const payloads = [];
    for (let k = 0; k < S; k++) payloads.push(s('async stream ' + k));
    const sigs = a.modulateData(payloads);
    const n = sigs.reduce((m, x) => Math.max(m, x.length), 0) + 64;
    const buf = new Float32Array(S * n);
    sigs.forEach((x, k) => buf.set(x, k * n));
    const want = a.demodulateData(Float32Array.from(buf), n, n, false);
    let ticks = 0;
    const timer = setInterval(() => { ticks++; }, 0);
    const pending = b.demodulateDataAsync(buf, n, n, false);
    await assert.rejects(b.demodulateDataAsync(buf, n, n, false), /already in flight/);
    const got = await pending;
    clearInterval(timer);
    for (let k = 0; k < S; k++) {
      assert.deepStrictEqual(Array.from(got.bytes[k]), Array.from(want.bytes[k]));
      assert.strictEqual(Buffer.from(got.bytes[k]).toString('ascii'), 'async stream ' + k);
    }
    a.close(); b.close();
  }
  // ---- FSKBatchSharded: three engines (all on device 0 here) driven together == one engine with all streams ----
  {
    const S = 50;
    const cfg = { baudRate: 1200, markFrequency: 1200, spaceFrequency: 2200 };
    const one = new M.FSKBatch(S, cfg), many = new M.FSKBatchSharded(S, cfg, { devices: [0, 0, 0] });
    assert.deepStrictEqual(many.shards.map((x) => [x.first, x.count]), [[0, 17], [17, 17], [34, 16]]);
    const payloads = [];
    for (let k = 0; k < S; k++) payloads.push(s('sharded stream ' + k));
    const sigs = many.modulateData(payloads);
    one.modulateData(payloads).forEach((x, k) => assert.deepStrictEqual(Array.from(x), Array.from(sigs[k])));
    const n = sigs.reduce((m, x) => Math.max(m, x.length), 0) + 64;
    const buf = new Float32Array(S * n);
    sigs.forEach((x, k) => buf.set(x, k * n));
    const want = one.demodulateData(Float32Array.from(buf), n, n, false);
    const got = await many.demodulateData(buf, n, n, false);
    for (let k = 0; k < S; k++) {
      assert.strictEqual(Buffer.from(got.bytes[k]).toString('ascii'), 'sharded stream ' + k);
      assert.deepStrictEqual(Array.from(got.bytes[k]), Array.from(want.bytes[k]));
    }
    assert.deepStrictEqual(Array.from(got.eod), Array.from(want.eod));
    assert.deepStrictEqual(many.getStatus(40), one.getStatus(40));
    many.reset(40); one.reset(40);
    assert.deepStrictEqual(many.getStatus(40), one.getStatus(40));
    assert.throws(() => many.getStatus(50), /out of range/);
    one.close(); many.close();
  }
  // ---- FIR (tests/dsp/filters.node.test.ts:190-206: impulse response = taps) ----
  {
    const taps = [0.1, -0.2, 0.3, 0.25, -0.05];
    const f = new F.FIRFilter(taps);
    const y = [];
    for (let i = 0; i < 8; i++) y.push(f.process(i === 0 ? 1 : 0));
    assert.deepStrictEqual(y, taps.map((t) => Math.fround(t)).concat([0, 0, 0]));
    f.reset();
    assert.strictEqual(f.process(1), Math.fround(0.1));
    f.close();
    const lp = F.FilterFactory.createFIRLowpass(1000, 48000);
    assert.strictEqual(lp.getCoefficients().length, 51);
    const dc = lp.processBuffer(new Float32Array(200).fill(1));
    const gain = lp.getCoefficients().reduce((a, b) => a + b, 0);
    assert.ok(Math.abs(dc[199] - gain) < 1e-6);       // DC gain = sum of taps
    lp.close();
  }
  // ---- IIR (src/dsp/filters.ts:8-106; tests/dsp/filters-advanced.node.test.ts:115-143) ----
  {
    // KATs of SURVEY.md section 8 (a4 / a5)
    const lpd = F.FilterDesign.butterworthLowpass(1200, 48000);
    assert.strictEqual(lpd.b[0], 0.005542717210280682); assert.strictEqual(lpd.a[1], -1.7786317778245848); assert.strictEqual(lpd.a[2], 0.8008026466657076);
    const f = new F.IIRFilter([2.0, 1.0, 0.5], [2.0, -0.5, 0.25]);          // a0 normalisation
    assert.deepStrictEqual(f.getCoefficients(), { b: [1.0, 0.5, 0.25], a: [1.0, -0.25, 0.125] });
    // the difference equation by hand, in doubles, in the reference's order
    const b = [1.0, 0.5, 0.25], a = [1.0, -0.25, 0.125], xs = [1, 0.5, -0.25, 0.125, 0, 0, 1, -1], want = [];
    let x1 = 0, x2 = 0, y1 = 0, y2 = 0;
    for (const x of xs) {
      let o = 0; o += b[0] * x; o += b[1] * x1; o += b[2] * x2; o -= a[1] * y1; o -= a[2] * y2;
      want.push(o); x2 = x1; x1 = x; y2 = y1; y1 = o;
    }
    assert.deepStrictEqual(xs.map((x) => f.process(x)), want);
    f.reset();
    assert.deepStrictEqual(Array.from(f.processBuffer(Float32Array.from(xs))), want.map((v) => Math.fround(v)));
    f.close();
    const lp = F.FilterFactory.createIIRLowpass(1200, 48000);
    const dc = lp.processBuffer(new Float32Array(2000).fill(1));
    assert.ok(Math.abs(dc[1999] - 1) < 1e-4);                                  // unity DC gain
    lp.close();
    const many = new F.IIRFilterBatch(lpd.b, lpd.a, 70);
    const blk = new Float32Array(70 * 64);
    for (let i = 0; i < blk.length; i++) blk[i] = Math.sin(i * 0.37) * ((i >> 6) + 1);
    const ym = many.processBuffer(blk);
    const one = new F.IIRFilter(lpd.b, lpd.a);
    assert.deepStrictEqual(Array.from(ym.subarray(69 * 64, 70 * 64)), Array.from(one.processBuffer(blk.subarray(69 * 64, 70 * 64))));
    one.close(); many.close();
  }
  console.log('js next gpu tests ok');
}

(mode === 'gpu' ? gpuTests() : cpuTests()).catch((e) => { console.error(e); process.exit(1); });
