"""FSKHIP_* environment variables -> fskhip_set_option() names (tests/conftest.py and the measurement tools install this as
webaudio_modem_amd.engine.option_hook; the package and libfskhip.so themselves read no environment variable).

  FSKHIP_SPLIT=0|1|4|5|6|a|b|c  kernel = one-wave | two-wave | four-wave | five-wave | seven-wave | auto | auto-r02 | auto-r04
  FSKHIP_FORCE_GENERIC=1      force_generic
  FSKHIP_BLK_YSLOTS=<n>       blk_y_slots          FSKHIP_BLK_MIN_TILES=<n>   blk_min_tiles
  FSKHIP_BLK_RESIDENT=<n>     blk_resident         FSKHIP_SLICE_TILES=<n>|off slice_tiles
  FSKHIP_HOST_SLAB=<n>        host_slab
  FSKHIP_SIX_MIN_TILES=<n>    stage_min_tiles      FSKHIP_SIX_YSLOTS=<n>      stage_y_slots    FSKHIP_SIX_ROLES=<7 digits> stage_roles
"""
import os

_KERNEL = {"0": "one-wave", "1": "two-wave", "4": "four-wave", "5": "five-wave", "6": "seven-wave", "a": "auto", "b": "auto-r02", "c": "auto-r04"}


def from_env(n_streams=None, precision=None):
    o = {}
    sp = os.environ.get("FSKHIP_SPLIT")
    if sp:
        o["kernel"] = _KERNEL[sp[0]]
    if os.environ.get("FSKHIP_FORCE_GENERIC", "")[:1] == "1":
        o["force_generic"] = 1
    for env, name in (("FSKHIP_BLK_YSLOTS", "blk_y_slots"), ("FSKHIP_BLK_MIN_TILES", "blk_min_tiles"),
                      ("FSKHIP_BLK_RESIDENT", "blk_resident"), ("FSKHIP_SLICE_TILES", "slice_tiles"),
                      ("FSKHIP_BLK_LANES", "blk_lanes"), ("FSKHIP_BLK_RESETS", "blk_resets"),
                      ("FSKHIP_SIX_MIN_TILES", "stage_min_tiles"), ("FSKHIP_SIX_YSLOTS", "stage_y_slots"), ("FSKHIP_SIX_ROLES", "stage_roles"),
                      ("FSKHIP_HOST_SLAB", "host_slab")):
        v = os.environ.get(env)
        if v is not None and v != "":
            o[name] = v
    return o


def install():
    import webaudio_modem_amd.engine as eng
    eng.option_hook = from_env
