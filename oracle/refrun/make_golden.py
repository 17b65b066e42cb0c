#!/usr/bin/env python3
"""Regenerate tests/golden/ from the REAL reference (build container only).

TEST INFRASTRUCTURE.  Steps:
  1. strip_ts.py type-strips /root/reference/src/{utils,dsp/filters,core,modems/fsk}.ts into a
     temp dir (never into the repo);
  2. golden_harness.js runs the reference FSKCore under Node on the scenario list and dumps raw
     arrays + a manifest into the same temp dir;
  3. this script packs the arrays into tests/golden/golden.npz (compressed) and writes
     tests/golden/manifest.json; golden_harness_next.js does the same for the SURVEY 8(f) rows (CRC-16 /
     XModem packets, ChunkedModulator, the FSKProcessor quantum loop) -> golden_next.npz + manifest_next.json;
     golden_harness_hostile.js for NaN / Inf / out-of-range / subnormal input (round 6) -> golden_hostile.npz + manifest_hostile.json.
Only data (inputs, expected outputs, status snapshots, intermediates) reaches the repo.

usage: python oracle/refrun/make_golden.py [--ref /root/reference]
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default="", help="'golden.npz', 'golden_next' or 'golden_hostile' to regenerate one set")
    ap.add_argument("--keep", action="store_true", help="keep the temp dir (debugging)")
    args = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix="fsk_golden_")
    try:
        subprocess.check_call([sys.executable, os.path.join(HERE, "strip_ts.py"), args.ref, tmp])
        gold = os.path.join(REPO, "tests", "golden")
        os.makedirs(gold, exist_ok=True)
        jobs = [("golden_harness.js", "golden.npz", "manifest.json"),
                ("golden_harness_next.js", "golden_next.npz", "manifest_next.json"),
                ("golden_harness_hostile.js", "golden_hostile.npz", "manifest_hostile.json")]
        for harness, npz, man_name in jobs:
            if args.only and args.only not in npz:
                continue
            out = os.path.join(tmp, "out_" + npz)
            subprocess.check_call(["node", os.path.join(HERE, harness), os.path.join(tmp, "ref_bundle.js"), out])
            with open(os.path.join(out, "manifest.json")) as fh:
                man = json.load(fh)
            arrays = {}
            for name, meta in man["arrays"].items():
                dt = {"f4": "<f4", "f8": "<f8", "u1": "u1", "i4": "<i4"}[meta["dtype"]]
                a = np.fromfile(os.path.join(out, "%s.%s.bin" % (name, meta["dtype"])), dtype=dt)
                assert a.size == meta["n"], name
                arrays[name] = a
            np.savez_compressed(os.path.join(gold, npz), **arrays)
            with open(os.path.join(gold, man_name), "w") as fh:
                json.dump(man, fh, indent=None, separators=(",", ":"))
            sz = os.path.getsize(os.path.join(gold, npz))
            print("%s: wrote %d arrays, %s %.2f MB" % (harness, len(arrays), npz, sz / 1e6))
    finally:
        if args.keep:
            print("kept", tmp)
        else:
            shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
