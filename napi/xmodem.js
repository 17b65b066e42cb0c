'use strict';
// CRC16 / XModemPacket / ControlType with the reference's surface (src/utils/crc16.ts, src/transports/xmodem/packet.ts,
// types.ts), plus the batch forms and scanBursts -- the receive checks of XModemTransport (xmodem.ts:233-320) applied
// to recorded bursts.  Everything that COMPUTES does so in libfskhip.so through the N-API addon (no JavaScript CRC here); the one
// method that only lays bytes out is XModemPacket.serialize(packet): packet.ts:44-54 writes the fields of the packet OBJECT as
// they are -- checksum included, whatever it holds (the reference's tests serialise packets with a wrong one) -- so it cannot go
// through serializeBatch, which computes the CRC.
const path = require('path');
const addon = require(path.join(__dirname, 'fsk_addon.node'));

const ControlType = Object.freeze({ SOH: 0x01, ACK: 0x06, NAK: 0x15, EOT: 0x04 });      // types.ts:29-34
const PacketConstants = Object.freeze({ SOH: 0x01, HEADER_SIZE: 4, CRC_SIZE: 2, MIN_PACKET_SIZE: 6, MAX_PACKET_SIZE: 261,
  MAX_PAYLOAD_SIZE: 255, MAX_SEQUENCE: 255, MIN_DATA_SEQUENCE: 1 });                       // types.ts:62-75
const XM_STATUS = ['need_more', 'eot', 'truncated', 'invalid_sequence', 'invalid_crc', 'unexpected_sequence'];
const XM_ERRORS = { 3: 'Invalid sequence number', 4: 'Invalid CRC', 5: 'Unexpected sequence number' };  // xmodem.ts:273,290,318

function packRows(rows) {
  let mx = 0;
  for (const r of rows) mx = Math.max(mx, r.length);
  const pitch = Math.max(4, (mx + 3) & ~3);
  const slab = new Uint8Array(pitch * rows.length);
  const lens = new Uint32Array(rows.length);
  rows.forEach((r, i) => { slab.set(r, i * pitch); lens[i] = r.length; });
  return { slab, pitch, lens };
}

function crc16Batch(rows, device = 0) {
  if (!rows.length) return new Uint16Array(0);
  const p = packRows(rows);
  return addon.crc16(p.slab, p.pitch, p.lens, device);
}

class CRC16 {                                   // crc16.ts:11-50
  static calculate(data, device = 0) { return crc16Batch([data], device)[0]; }
  static verify(data, expectedCrc, device = 0) { return CRC16.calculate(data, device) === expectedCrc; }
}

function serializeBatch(seqs, payloads, device = 0) {
  seqs.forEach((sequence, i) => {               // createData's throws (packet.ts:22-27)
    if (sequence < 1 || sequence > 255) throw new Error(`Invalid sequence: ${sequence}. Must be 1-255.`);
    if (payloads[i].length > 255) throw new Error(`Payload too large: ${payloads[i].length}. Max 255 bytes.`);
  });
  if (!seqs.length) return [];
  const p = packRows(payloads);
  const r = addon.xmodemSerialize(p.slab, p.pitch, p.lens, Uint32Array.from(seqs), device);
  return payloads.map((_, i) => r.out.slice(i * r.outPitch, i * r.outPitch + r.lens[i]));
}

class XModemPacket {                            // packet.ts:17-66
  static createData(sequence, payload, device = 0) {
    const wire = serializeBatch([sequence], [payload], device)[0];
    return { soh: wire[0], sequence: wire[1], invSequence: wire[2], length: wire[3], payload: new Uint8Array(payload),
      checksum: (wire[wire.length - 2] << 8) | wire[wire.length - 1] };
  }
  static serialize(packet) {
    const result = new Uint8Array(4 + packet.payload.length + 2);
    result[0] = packet.soh; result[1] = packet.sequence; result[2] = packet.invSequence; result[3] = packet.length;
    result.set(packet.payload, 4);
    result[4 + packet.payload.length] = (packet.checksum >> 8) & 0xFF;
    result[4 + packet.payload.length + 1] = packet.checksum & 0xFF;
    return result;
  }
  static verify(packet, device = 0) { return CRC16.calculate(packet.payload, device) === packet.checksum; }
  static serializeControl(controlType) { return new Uint8Array([controlType]); }
}

// bursts: array of Uint8Array (what the demodulator returned per stream); expected: starting expectedSequence per stream
function scanBursts(bursts, expected, device = 0) {
  if (!bursts.length) return [];
  const p = packRows(bursts);
  const exp = Uint32Array.from(bursts.map((_, i) => (Array.isArray(expected) || ArrayBuffer.isView(expected)) ? expected[i] : expected));
  const r = addon.xmodemScan(p.slab, p.pitch, p.lens, exp, device);
  return bursts.map((_, i) => {
    const q = r.results.subarray(i * 10, i * 10 + 10);
    return { status: q[0], statusName: XM_STATUS[q[0]], error: XM_ERRORS[q[0]] || null, expectedAfter: q[1], packets: q[2], dropped: q[3],
      consumed: q[4], errSeq: q[6], errLen: q[7], crcRx: q[8], crcCalc: q[9], data: r.data.slice(i * r.dataPitch, i * r.dataPitch + q[5]) };
  });
}

module.exports = { CRC16, XModemPacket, ControlType, PacketConstants, crc16Batch, serializeBatch, scanBursts };
