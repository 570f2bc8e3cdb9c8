#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE ITSELF.

Runs only in the build container: it needs oracle/_ref/libhrfd_ref.so, which
oracle/Makefile compiles from the reference's own sources under /root/reference.
The fixtures hold inputs (by generator name + seed + sha256) and the
reference's outputs; no reference source text is stored.

    python tests/golden/make_golden.py
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
from tests.reflib import AM, FM, WBFM, LSB, USB, MODE_NAMES  # noqa: E402

BLK = synth.BLOCK_BYTES
N_BLOCKS = 8
KINDS = ["lcg", "fmtone", "amtone", "dc_pos", "dc_neg", "impulse", "zeros"]


def main():
    ref = reflib.Ref()
    arrays = {}
    manifest = {"rx": [], "frontend": [], "rx_long": [], "squelch": [], "chunked": [], "tx": [],
                "interp": [], "nco": [], "tables": {}}

    # --- rx: per mode x input kind, 8 blocks: PCM + per-block magnitude
    for mode in (AM, FM, WBFM, LSB, USB):
        for kind in KINDS:
            seed = 1 if kind == "lcg" else 2
            x = synth.make_input(kind, seed, N_BLOCKS)
            h = ref.rx()
            h.set_mode(mode)
            pcm, mags = [], []
            for b in range(N_BLOCKS):
                p, m, _, _ = h.process(x[b * BLK:(b + 1) * BLK])
                assert len(p) == 512
                pcm.append(p)
                mags.append(m)
            key = f"rx_{MODE_NAMES[mode]}_{kind}"
            arrays[key + "_pcm"] = np.concatenate(pcm)
            arrays[key + "_mag"] = np.array(mags, dtype=np.uint32)
            manifest["rx"].append({"key": key, "mode": mode, "kind": kind, "seed": seed,
                                   "blocks": N_BLOCKS, "input_sha256": synth.digest(x)})

    # --- front end only: decimatedData (after the Fs/4 mix) of block 0 and 1
    for kind, seed in (("lcg", 1), ("lcg", 77), ("fmtone", 5), ("dc_pos", 0), ("dc_neg", 0)):
        x = synth.make_input(kind, seed, 2)
        h = ref.rx()
        outs = [h.process(x[b * BLK:(b + 1) * BLK])[3] for b in range(2)]
        key = f"fe_{kind}_{seed}"
        arrays[key] = np.concatenate(outs)
        manifest["frontend"].append({"key": key, "kind": kind, "seed": seed, "blocks": 2,
                                     "input_sha256": synth.digest(x)})

    # --- long runs: sha256 of 64 blocks of PCM
    for mode in (AM, FM, WBFM, LSB, USB):
        x = synth.make_input("lcg", 100 + mode, 64)
        h = ref.rx()
        h.set_mode(mode)
        pcm = np.concatenate([h.process(x[b * BLK:(b + 1) * BLK])[0] for b in range(64)])
        manifest["rx_long"].append({"mode": mode, "kind": "lcg", "seed": 100 + mode, "blocks": 64,
                                    "pcm_sha256": synth.digest(pcm), "input_sha256": synth.digest(x)})

    # --- squelch: threshold -30 dBFS, loud/quiet pattern; PCM lengths + PCM
    pattern = [1, 1, 0, 0, 1, 0, 0, 0, 1]
    loud = synth.make_input("fmtone", 1, 1)
    quiet = synth.zeros_iq(synth.BLOCK_IQ)
    h = ref.rx()
    h.set_mode(WBFM)
    h.set_threshold(-30)
    lens, pcm, mags = [], [], []
    for bit in pattern:
        p, m, _, _ = h.process(loud if bit else quiet)
        lens.append(len(p)); pcm.append(p); mags.append(m)
    arrays["squelch_pcm"] = np.concatenate(pcm)
    manifest["squelch"].append({"pattern": pattern, "threshold": -30, "lens": lens, "mags": mags,
                                "key": "squelch_pcm"})

    # --- chunk invariance: 64-byte calls
    x = synth.make_input("fmtone", 3, 1)
    for mode in (AM, FM, WBFM, LSB, USB):
        h = ref.rx()
        h.set_mode(mode)
        parts = [h.process(x[o:o + 64])[0] for o in range(0, len(x), 64)]
        key = f"chunk64_{MODE_NAMES[mode]}"
        arrays[key] = np.concatenate(parts)
        manifest["chunked"].append({"key": key, "mode": mode, "kind": "fmtone", "seed": 3, "chunk": 64})

    # --- gain overflow quirk (float -> int16 wrap), 2 blocks
    for mode, gain in ((WBFM, 1e6), (AM, 1e12), (LSB, 5e4), (FM, 123456.0)):
        x = synth.make_input("lcg", 11, 2)
        h = ref.rx()
        h.set_mode(mode)
        h.set_gain(mode, gain)
        key = f"gain_{MODE_NAMES[mode]}"
        arrays[key] = np.concatenate([h.process(x[b * BLK:(b + 1) * BLK])[0] for b in range(2)])
        manifest["rx"].append({"key": key, "mode": mode, "kind": "lcg", "seed": 11, "blocks": 2,
                               "gain": gain, "gain_case": True})

    # --- tx: SSB modulator, first 4 KiB of block 0 + sha256 of 8 blocks
    for lsb in (1, 0):
        pcm_in = synth.lcg_pcm(7, 8 * 512)
        m = ref.ssbmod(bool(lsb))
        out = np.concatenate([m.process(pcm_in[b * 512:(b + 1) * 512]) for b in range(8)])
        key = f"ssbmod_{'lsb' if lsb else 'usb'}"
        arrays[key + "_head"] = out[:4096]
        arrays[key + "_tail"] = out[-4096:]
        manifest["tx"].append({"key": key, "lsb": lsb, "seed": 7, "blocks": 8, "iq_sha256": synth.digest(out)})

    # --- signals/interpolateSignal on an am.cc-style baseband (int16 IQ pairs)
    n = 2000
    t = np.arange(n)
    i16 = np.rint(12000 * (1 + 0.5 * np.sin(2 * np.pi * 400 * t / 8000.0))).astype(np.int16)
    q16 = np.zeros(n, dtype=np.int16)
    iq16 = np.empty(2 * n, dtype=np.int16); iq16[0::2] = i16; iq16[1::2] = q16
    out = ref.interpolate_signal(iq16)
    arrays["interp_in"] = iq16
    arrays["interp_head"] = out[:8192]
    manifest["interp"].append({"pairs": n, "iq_sha256": synth.digest(out)})
    iq16 = synth.lcg_pcm(9, 2 * 600)
    out = ref.interpolate_signal(iq16)
    manifest["interp"].append({"pairs": 600, "lcg_seed": 9, "iq_sha256": synth.digest(out)})

    # --- Nco: 1000 samples per mode + table edges
    for fs, f in ((8000.0, 1000.0), (256000.0, 75000.0)):
        nco = ref.nco(fs, f)
        s, c = nco.tables()
        i0, q0 = nco.run(1000, False)
        i1, q1 = nco.run(1000, True)
        key = f"nco_{int(fs)}_{int(f)}"
        arrays[key + "_run"] = np.stack([i0, q0])
        arrays[key + "_fast"] = np.stack([i1, q1])
        manifest["nco"].append({"key": key, "fs": fs, "f": f, "sin_sha256": synth.digest(s),
                                "cos_sha256": synth.digest(c)})
    arrays["nco_sin_edges"] = np.concatenate([s[:8], s[8188:8196], s[-8:]])
    arrays["nco_cos_edges"] = np.concatenate([c[:8], c[8188:8196], c[-8:]])

    # --- tables
    arrays["dbfs_table"] = ref.dbfs_table()
    orc = reflib.Oracle()
    for name in ["HB1", "HB2", "HB3", "WBFM_D1", "POST_D12", "AUDIO_D40", "FM_TUNER_D32", "AM_D1", "AM_D2",
                 "AM_D3", "SSB_DELAY", "SSB_HILBERT", "INTERP_HB8", "INTERP_HB3", "INTERP_HB2", "INTERP_HB1",
                 "INTERPSIG_S1"]:
        manifest["tables"][name] = ref.quantise(orc.table(name)).tolist()

    np.savez_compressed(os.path.join(HERE, "golden.npz"), **arrays)
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    size = os.path.getsize(os.path.join(HERE, "golden.npz"))
    print(f"wrote golden.npz ({size} bytes, {len(arrays)} arrays) and golden.json")


if __name__ == "__main__":
    main()
