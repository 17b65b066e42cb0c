"""CPU oracle for the SURVEY.md 8(f) rows: CRC-16 / XModem packets, ChunkedModulator, the FSKProcessor quantum loop.

TEST INFRASTRUCTURE.  Only tests/ and __graft_entry__.smoke() may import this module; nothing under
webaudio_modem_amd/ or napi/ does.  Plain Python restatements (sizes here are small), each citing the reference
file:line it follows; pinned against tests/golden/golden_next.npz (outputs of the real reference classes under
Node, oracle/refrun/golden_harness_next.js) by tests/test_oracle_next.py.
"""
import numpy as np

SOH, ACK, NAK, EOT = 0x01, 0x06, 0x15, 0x04  # types.ts:29-34

# scan status codes, shared with include/fskhip.h (FSKHIP_XM_*)
XM_NEED_MORE, XM_EOT, XM_TRUNCATED, XM_INVALID_SEQUENCE, XM_INVALID_CRC, XM_UNEXPECTED_SEQUENCE = 0, 1, 2, 3, 4, 5
XM_NAMES = {XM_NEED_MORE: "need_more", XM_EOT: "eot", XM_TRUNCATED: "truncated", XM_INVALID_SEQUENCE: "invalid_sequence",
            XM_INVALID_CRC: "invalid_crc", XM_UNEXPECTED_SEQUENCE: "unexpected_sequence"}


def crc16(data):
    """crc16.ts:21-38: CRC-16-CCITT, poly 0x1021, init 0xFFFF, no final xor, MSB first."""
    crc = 0xFFFF
    for byte in bytes(data):
        crc ^= byte << 8
        for _ in range(8):
            if crc & 0x8000:
                crc = (crc << 1) ^ 0x1021
            else:
                crc <<= 1
            crc &= 0xFFFF
    return crc


def create_data(sequence, payload):
    """packet.ts:21-39 (same error texts)."""
    payload = bytes(payload)
    if sequence < 1 or sequence > 255:
        raise ValueError("Invalid sequence: %d. Must be 1-255." % sequence)
    if len(payload) > 255:
        raise ValueError("Payload too large: %d. Max 255 bytes." % len(payload))
    return {"soh": SOH, "sequence": sequence, "invSequence": (~sequence) & 0xFF, "length": len(payload),
            "payload": payload, "checksum": crc16(payload)}


def serialize(packet):
    """packet.ts:44-54."""
    return bytes([packet["soh"], packet["sequence"], packet["invSequence"], packet["length"]]) + packet["payload"] + \
        bytes([(packet["checksum"] >> 8) & 0xFF, packet["checksum"] & 0xFF])


def scan_burst(data, expected):
    """The receive grammar of XModemTransport over a recorded burst: xmodem.ts:233-320 (receiveAllPackets,
    receiveAndProcessPacket, isPreviousSequence 525-530, assembleData 322-333) with "no more bytes" where the
    reference would time out.  Stops at the first error, like the reference's throw with maxRetries 0.  Pinned to the
    REAL class (tests/golden/manifest_next.json "scans": receiveData() over a scripted data channel): `packets` is
    statistics.packetsReceived (counted when the payload has been read, before its CRC is checked, xmodem.ts:280),
    `consumed` the bytes taken out of the receive buffer (a time-out after SOH leaves SOH consumed, after the 3-byte header
    the header too: waitForBytes, xmodem.ts:475-499)."""
    data = bytes(data)
    pos = 0
    out = bytearray()
    r = dict(status=XM_NEED_MORE, packets=0, dropped=0, consumed=0, err_seq=-1, err_len=-1, crc_rx=-1, crc_calc=-1)
    while True:
        if pos >= len(data):
            r["consumed"] = pos
            break
        first = data[pos]
        if first == EOT:
            r["status"] = XM_EOT
            r["consumed"] = pos + 1
            break
        if first != SOH:
            pos += 1
            continue
        if pos + 4 > len(data):
            r["status"] = XM_TRUNCATED
            r["consumed"] = pos + 1
            break
        seq, nseq, ln = data[pos + 1], data[pos + 2], data[pos + 3]
        if seq + nseq != 255:
            r.update(status=XM_INVALID_SEQUENCE, err_seq=seq, err_len=ln, consumed=pos + 4)
            r["dropped"] += 1
            break
        prev = 255 if expected == 1 else expected - 1
        if seq != expected and seq != prev:
            r.update(status=XM_UNEXPECTED_SEQUENCE, err_seq=seq, err_len=ln, consumed=pos + 4)
            r["dropped"] += 1
            break
        if pos + 6 + ln > len(data):
            r.update(status=XM_TRUNCATED, err_seq=seq, err_len=ln, consumed=pos + 4)
            break
        if seq == expected:
            payload = data[pos + 4:pos + 4 + ln]
            crc = (data[pos + 4 + ln] << 8) | data[pos + 5 + ln]
            calc = crc16(payload)
            r["packets"] += 1
            if calc != crc:
                r.update(status=XM_INVALID_CRC, err_seq=seq, err_len=ln, crc_rx=crc, crc_calc=calc, consumed=pos + 6 + ln)
                r["dropped"] += 1
                break
            out += payload
            expected = (expected % 255) + 1
        else:
            r["dropped"] += 1
        pos += 6 + ln
    r["expected_after"] = expected
    r["data"] = bytes(out)
    return r


class ByteRing:
    """RingBuffer(Uint8Array, n) with an integral capacity: utils.ts:38-58 (put overwrites the oldest)."""

    def __init__(self, cap):
        self.cap = cap
        self.buf = bytearray(cap)
        self.w = self.r = self.length = 0

    def put(self, v):
        self.buf[self.w] = v
        self.w = (self.w + 1) % self.cap
        if self.length < self.cap:
            self.length += 1
        else:
            self.r = (self.r + 1) % self.cap

    def remove(self):
        if self.length == 0:
            raise IndexError("Buffer is empty")
        v = self.buf[self.r]
        self.r = (self.r + 1) % self.cap
        self.length -= 1
        return v

    def clear(self):
        self.w = self.r = self.length = 0

    def drain(self):
        return bytes(self.remove() for _ in range(self.length))


class ChunkedModulator:
    """chunked-modulator.ts:22-88 over any object with modulate(bytes) -> float32 array."""

    def __init__(self, modulator):
        self.modulator = modulator
        self.pending = None
        self.pos = 0

    def start_modulation(self, data):
        if len(data) == 0:
            self._reset()
            return
        self.pending = np.asarray(self.modulator.modulate(bytes(data)), dtype=np.float32)
        self.pos = 0

    def get_next_samples(self, count):
        if self.pending is None:
            return None
        remaining = len(self.pending) - self.pos
        if remaining <= 0:
            return None
        n = min(count, remaining)
        signal = self.pending[self.pos:self.pos + n].copy()
        self.pos += n
        total = len(self.pending)
        if self.pos >= total:
            self._reset()
            return dict(signal=signal, isComplete=True, samplesConsumed=total, totalSamples=total)
        return dict(signal=signal, isComplete=False, samplesConsumed=self.pos, totalSamples=total)

    def is_modulating(self):
        return self.pending is not None

    def get_progress(self):
        return self.pos / len(self.pending) if self.pending is not None else 0

    def cancel(self):
        self._reset()

    def _reset(self):
        self.pending = None
        self.pos = 0


class ProcessorOracle:
    """One FSKProcessor: process() per quantum (fsk-processor.ts:152-167), modulateTo 256-276, processDemodulation
    294-322, demodulate() 117-138, the 'modulate' handler's RX clear on completion 228-235."""

    def __init__(self, core, rx_capacity=1024, clear_rx_on_tx_complete=True):
        self.core = core  # oracle.pyoracle.OracleCore
        self.ring = ByteRing(rx_capacity)
        self.pending = None
        self.completed = 0
        self.clear_rx = clear_rx_on_tx_complete

    def modulate(self, data):
        if self.pending is not None:
            raise RuntimeError("Modulation already in progress")
        self.pending = ChunkedModulator(self.core)
        self.pending.start_modulation(data)

    def process(self, inp, n_out):
        if inp is not None:
            got, _ = self.core.demodulate(np.asarray(inp, dtype=np.float32))
            for b in got:
                self.ring.put(b)
        out = np.zeros(n_out, dtype=np.float32)
        if self.pending is not None:
            r = self.pending.get_next_samples(n_out)
            if r is not None:
                out[:len(r["signal"])] = r["signal"]
                if r["isComplete"]:
                    self.pending = None
                    self.completed += 1
                    if self.clear_rx:
                        self.ring.clear()
        return out

    def demodulate(self):
        return self.ring.drain()
