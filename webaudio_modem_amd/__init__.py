"""webaudio_modem_amd: MI355X-native batch FSK DSP engine behind the reference's FSKCore surface.

Everything that computes goes through libfskhip.so (hand-written HIP for gfx950, C ABI in
include/fskhip.h).  There is no CPU path in this package.
"""
from ._lib import FskHipError, PRECISION_F32, PRECISION_F64, DEMOD_WRITEBACK_AGC, LIB_PATH  # noqa: F401
from .engine import FSKEngine, DEFAULT_FSK_CONFIG, make_config, pinned_empty  # noqa: F401
from .fsk_core import FSKCore, Event, EventEmitter  # noqa: F401
from .filters import FilterDesign, FilterFactory, FIRFilter, FIRFilterBatch, IIRFilter, IIRFilterBatch  # noqa: F401
from .processor import ChunkedModulator, FSKProcessorBatch  # noqa: F401
from .xmodem import CRC16, XModemPacket, ControlType, crc16_batch, serialize_batch, scan_bursts  # noqa: F401
from . import sharding  # noqa: F401
from .sharded import FSKEngineSharded  # noqa: F401

__all__ = ["FSKEngine", "FSKEngineSharded", "FSKCore", "FilterDesign", "FilterFactory", "FIRFilter", "FIRFilterBatch", "IIRFilter", "IIRFilterBatch", "ChunkedModulator",
           "FSKProcessorBatch", "CRC16", "XModemPacket", "ControlType", "crc16_batch", "serialize_batch", "scan_bursts",
           "DEFAULT_FSK_CONFIG", "FskHipError", "PRECISION_F32", "PRECISION_F64"]
