#!/usr/bin/env python3
"""Golden vectors for the AM / FM / WBFM modulators, produced by the REFERENCE's own
compiled AmModulator / FmModulator (oracle/_ref/libhrfd_ref.so, built by oracle/Makefile from
/root/reference).  Build container only.  Inputs are named generators + seeds; outputs are the
reference's int8 IQ: head, tail and sha256 of 4 calls of 512 PCM samples for AM and WBFM
(bit-exact paths), and the complete output of 2 calls of 64 samples for FM (float trig path, compared
within +-1 LSB on the device, so a hash is of no use there).

    python tests/golden/make_golden_mod.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from hackrfdiags_amd import synth  # noqa: E402
from tests import reflib  # noqa: E402


def main():
    ref = reflib.Ref()
    arrays, manifest = {}, {"am": [], "fm": [], "wbfm": []}
    for idx in (None, 0.5):
        pcm = synth.lcg_pcm(21, 4 * 512)
        m = ref.ammod()
        if idx is not None:
            m.set_param(idx)
        out = m.process(pcm)
        key = "ammod_default" if idx is None else "ammod_idx050"
        arrays[key + "_head"] = out[:4096]
        arrays[key + "_tail"] = out[-4096:]
        manifest["am"].append({"key": key, "index": idx, "seed": 21, "calls": 4, "iq_sha256": synth.digest(out)})
    for dev in (None, 1000.0):
        pcm = synth.lcg_pcm(22, 128)
        m = ref.fmmod()
        if dev is not None:
            m.set_param(dev)
        out = np.concatenate([m.process(pcm[:64]), m.process(pcm[64:])])
        key = "fmmod_default" if dev is None else "fmmod_dev1000"
        arrays[key] = out
        manifest["fm"].append({"key": key, "deviation": dev, "seed": 22, "calls": [64, 64], "iq_sha256": synth.digest(out)})
    for dev in (None, 40000.0):
        pcm = synth.lcg_pcm(23, 4 * 512)
        m = ref.wbfmmod()
        if dev is not None:
            m.set_param(dev)
        out = m.process(pcm)
        key = "wbfmmod_default" if dev is None else "wbfmmod_dev40000"
        arrays[key + "_head"] = out[:4096]
        arrays[key + "_tail"] = out[-4096:]
        manifest["wbfm"].append({"key": key, "deviation": dev, "seed": 23, "calls": 4, "iq_sha256": synth.digest(out)})
    np.savez_compressed(os.path.join(HERE, "golden_mod.npz"), **arrays)
    with open(os.path.join(HERE, "golden_mod.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print({k: v.shape for k, v in arrays.items()})


if __name__ == "__main__":
    main()
