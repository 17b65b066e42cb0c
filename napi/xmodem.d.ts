// Typings of napi/xmodem.js: CRC16 / XModemPacket / ControlType with the reference's surface (src/utils/crc16.ts,
// src/transports/xmodem/packet.ts, types.ts) plus the batch forms and the receive-grammar scan.
export declare const ControlType: Readonly<{ SOH: 0x01; ACK: 0x06; NAK: 0x15; EOT: 0x04 }>;
export declare const PacketConstants: Readonly<{ SOH: 0x01; HEADER_SIZE: 4; CRC_SIZE: 2; MIN_PACKET_SIZE: 6; MAX_PACKET_SIZE: 261;
  MAX_PAYLOAD_SIZE: 255; MAX_SEQUENCE: 255; MIN_DATA_SEQUENCE: 1 }>;
export interface DataPacket { soh: number; sequence: number; invSequence: number; length: number; payload: Uint8Array; checksum: number; }
export declare class CRC16 {
  static calculate(data: Uint8Array, device?: number): number;
  static verify(data: Uint8Array, expectedCrc: number, device?: number): boolean;
}
export declare class XModemPacket {
  static createData(sequence: number, payload: Uint8Array, device?: number): DataPacket;
  static serialize(packet: DataPacket): Uint8Array;
  static verify(packet: DataPacket, device?: number): boolean;
  static serializeControl(controlType: number): Uint8Array;
}
export declare function crc16Batch(rows: Uint8Array[], device?: number): Uint16Array;
export declare function serializeBatch(seqs: number[], payloads: Uint8Array[], device?: number): Uint8Array[];
export interface ScanResult {
  status: number; statusName: 'need_more' | 'eot' | 'truncated' | 'invalid_sequence' | 'invalid_crc' | 'unexpected_sequence';
  /** the reference's exception text where XModemTransport would throw (xmodem.ts:273, 290, 318), else null */
  error: string | null;
  expectedAfter: number; packets: number; dropped: number; consumed: number; errSeq: number; errLen: number; crcRx: number; crcCalc: number;
  data: Uint8Array;
}
export declare function scanBursts(bursts: Uint8Array[], expected: number | ArrayLike<number>, device?: number): ScanResult[];
