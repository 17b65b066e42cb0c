"""CRC-16 and XModem packets above the C ABI (include/fskhip_next.h): the reference's `CRC16`
(src/utils/crc16.ts) and `XModemPacket` (src/transports/xmodem/packet.ts) with the same names, argument meaning
and error texts, plus the batch forms that actually feed a GPU and `scan_bursts`, the receive checks of
XModemTransport (src/transports/xmodem/xmodem.ts:233-320) applied to the bytes a demodulate call returned.
Everything computes in libfskhip.so; there is no CPU path here.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import XModemResult, XM_NEED_MORE, XM_EOT, XM_TRUNCATED, XM_INVALID_SEQUENCE, XM_INVALID_CRC, \
    XM_UNEXPECTED_SEQUENCE  # noqa: F401


class ControlType:  # types.ts:29-34
    SOH, ACK, NAK, EOT = 0x01, 0x06, 0x15, 0x04


PacketConstants = dict(SOH=0x01, HEADER_SIZE=4, CRC_SIZE=2, MIN_PACKET_SIZE=6, MAX_PACKET_SIZE=261,
                       MAX_PAYLOAD_SIZE=255, MAX_SEQUENCE=255, MIN_DATA_SEQUENCE=1)  # types.ts:62-75

STATUS_NAMES = {XM_NEED_MORE: "need_more", XM_EOT: "eot", XM_TRUNCATED: "truncated",
                XM_INVALID_SEQUENCE: "invalid_sequence", XM_INVALID_CRC: "invalid_crc",
                XM_UNEXPECTED_SEQUENCE: "unexpected_sequence"}
# what XModemTransport throws where a scan ends with that status (xmodem.ts:273, 290, 318)
STATUS_ERRORS = {XM_INVALID_SEQUENCE: "Invalid sequence number", XM_INVALID_CRC: "Invalid CRC",
                 XM_UNEXPECTED_SEQUENCE: "Unexpected sequence number"}


def _pack_rows(rows, min_pitch=4):
    """list of bytes-like -> (uint8 [n][pitch] array, uint32 lens)."""
    rows = [bytes(r) for r in rows]
    lens = np.array([len(r) for r in rows], dtype=np.uint32)
    pitch = max(min_pitch, (int(lens.max()) + 3) // 4 * 4 if len(rows) else min_pitch)
    slab = np.zeros((len(rows), pitch), dtype=np.uint8)
    for i, r in enumerate(rows):
        slab[i, :len(r)] = np.frombuffer(r, dtype=np.uint8)
    return slab, lens


def crc16_batch(rows, device=0):
    """CRC16.calculate of every row; rows is a list of bytes-like or (uint8 [n][pitch], lens)."""
    slab, lens = rows if isinstance(rows, tuple) else _pack_rows(rows)
    slab = np.ascontiguousarray(slab, dtype=np.uint8)
    lens = np.ascontiguousarray(lens, dtype=np.uint32)
    out = np.zeros(len(lens), dtype=np.uint16)
    _lib.check(_lib.lib().fskhip_crc16_host(device, slab.ctypes.data, slab.shape[1], lens.ctypes.data, len(lens),
                                            out.ctypes.data))
    return out


class CRC16:
    """crc16.ts: CRC-16-CCITT, polynomial 0x1021, initial value 0xFFFF, no final xor."""
    POLYNOMIAL, INITIAL_VALUE, FINAL_XOR = 0x1021, 0xFFFF, 0x0000

    @staticmethod
    def calculate(data, device=0):
        return int(crc16_batch([data], device)[0])

    @staticmethod
    def verify(data, expectedCrc, device=0):
        return CRC16.calculate(data, device) == expectedCrc


def serialize_batch(seqs, payloads, device=0):
    """XModemPacket.serialize(createData(seq, payload)) per row -> list of bytes."""
    rows = [bytes(p) for p in payloads]
    for seq, p in zip(seqs, rows):  # createData's throws, before anything is sent to the device
        if seq < 1 or seq > 255:
            raise ValueError("Invalid sequence: %d. Must be 1-255." % seq)
        if len(p) > 255:
            raise ValueError("Payload too large: %d. Max 255 bytes." % len(p))
    slab, lens = _pack_rows(rows)
    seqs = np.ascontiguousarray(seqs, dtype=np.uint32)
    out_pitch = int(lens.max()) + 6 if len(rows) else 6
    out = np.zeros((len(rows), out_pitch), dtype=np.uint8)
    out_lens = np.zeros(len(rows), dtype=np.uint32)
    _lib.check(_lib.lib().fskhip_xmodem_serialize_host(device, slab.ctypes.data, slab.shape[1], lens.ctypes.data,
                                                       seqs.ctypes.data, len(rows), out.ctypes.data, out_pitch,
                                                       out_lens.ctypes.data))
    return [out[i, :out_lens[i]].tobytes() for i in range(len(rows))]


class XModemPacket:
    """packet.ts:17-66, one packet at a time (a batch of one on the device)."""

    @staticmethod
    def createData(sequence, payload, device=0):
        payload = bytes(payload)
        wire = serialize_batch([sequence], [payload], device)[0]
        return {"soh": wire[0], "sequence": wire[1], "invSequence": wire[2], "length": wire[3],
                "payload": payload, "checksum": (wire[-2] << 8) | wire[-1], "_wire": wire}

    @staticmethod
    def serialize(packet):
        if "_wire" in packet and packet["_wire"][4:-2] == bytes(packet["payload"]):
            return packet["_wire"]
        return bytes([packet["soh"], packet["sequence"], packet["invSequence"], packet["length"]]) + \
            bytes(packet["payload"]) + bytes([(packet["checksum"] >> 8) & 0xFF, packet["checksum"] & 0xFF])

    @staticmethod
    def verify(packet, device=0):
        return CRC16.calculate(packet["payload"], device) == packet["checksum"]

    @staticmethod
    def serializeControl(controlType):
        return bytes([controlType])


def scan_bursts(bursts, expected, device=0, data_pitch=None):
    """The receive grammar over one recorded burst per stream.  bursts: list of bytes-like (or (slab, counts));
    expected: the starting expectedSequence per stream.  Returns a list of dicts: status (XM_*), status_name,
    error (the reference's exception text or None), expected_after, packets, dropped, consumed, err_seq, err_len,
    crc_rx, crc_calc, data (assembled payload bytes)."""
    slab, counts = bursts if isinstance(bursts, tuple) else _pack_rows(bursts)
    slab = np.ascontiguousarray(slab, dtype=np.uint8)
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    n = len(counts)
    exp = np.ascontiguousarray(np.broadcast_to(np.asarray(expected, dtype=np.uint32), (n,)))
    if data_pitch is None:
        data_pitch = max(4, slab.shape[1])
    data = np.zeros((n, data_pitch), dtype=np.uint8)
    res = (XModemResult * max(1, n))()
    _lib.check(_lib.lib().fskhip_xmodem_scan_host(device, slab.ctypes.data, slab.shape[1], counts.ctypes.data,
                                                  exp.ctypes.data, n, data.ctypes.data, data_pitch, res))
    out = []
    for i in range(n):
        r = res[i]
        d = {k: int(getattr(r, k)) for k, _ in XModemResult._fields_}
        d["status_name"] = STATUS_NAMES[d["status"]]
        d["error"] = STATUS_ERRORS.get(d["status"])
        d["data"] = data[i, :d["data_len"]].tobytes()
        out.append(d)
    return out
