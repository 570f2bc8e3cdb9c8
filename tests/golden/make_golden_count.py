#!/usr/bin/env python3
"""Golden vectors from the reference's OWN test input: signals/count.raw (five seconds of 8 kS/s int16 PCM, 80 000
bytes -- the data file signals/makeThem.sh and generateBaseband.sh feed through `./a.out < count.raw |
./interpolateSignal > $1.iq`, and the kind of file README.txt:117-136 says the author runs through the am / fm / wbfm /
ssb harnesses to make test vectors for the demodulators).  Produced by the REFERENCE's own sources compiled by
oracle/Makefile (build container only):

  makethem   signals/{am,dsb,pm,fm}.cc < count.raw | signals/interpolateSignal.cc        -> int8 IQ at 2.048 MS/s
  modulator  {Ssb,Am,Fm,WbFm}Modulator::acceptData over count.raw, 512 samples per call  -> int8 IQ
             (and the reference's own harness programs ssb.cc / am.cc / fm.cc / wbfm.cc < count.raw: the same bytes)
  loop       that IQ, moved down by 64 kHz (the radio tunes 64 kHz high and IqDataProcessor::upconvertByFsOver4 brings
             the signal back: Radio.cc:1187-1191 -- tests/toolsupport.retune_minus_64k is that channel), 262144 bytes at a
             time through IqDataProcessor::acceptIqData in the matching demodulator mode (LSB / AM / FM / WBFM)
                                                                                          -> int16 PCM at 8 kS/s
             -- the closed loop the author used the modulators for (README.txt:133-136); the recovered audio
             correlates with count.raw (the manifest says how well)

The fixture holds the input file itself (data of the reference's own tests), sha256 digests of every output, the first
and last 2 KiB of every IQ stream and the recovered PCM of every loop.  No reference source text is stored.

    python tests/golden/make_golden_count.py
"""
import json
import os
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from hackrfdiags_amd import synth  # noqa: E402
from tests import reflib, toolsupport as T  # noqa: E402
from tests.reflib import AM, FM, WBFM, LSB  # noqa: E402

SRC = "/root/reference/signals/count.raw"
BLK = synth.BLOCK_BYTES


def modulate(mod, pcm):
    return np.concatenate([mod.process(pcm[s:s + 512]) for s in range(0, len(pcm), 512)])


def demodulate(rx, iq):
    """whole 262144-byte blocks, then the rest (a multiple of 64 bytes: the reference is chunk-invariant there)"""
    out = []
    for s in range(0, len(iq), BLK):
        out.append(rx.process(iq[s:s + BLK])[0])
    return np.concatenate(out)


def main():
    shutil.copyfile(SRC, os.path.join(HERE, "count.raw"))
    pcm = np.fromfile(SRC, dtype="<i2")
    assert pcm.size == 40000
    ref = reflib.Ref()
    arrays, manifest = {}, {"input": {"file": "count.raw", "samples": int(pcm.size), "sha256": synth.digest(pcm)},
                            "makethem": [], "modulator": [], "loop": []}
    for kind in T.SIG_KINDS:
        iq = T.ref_interpolate(T.ref_siggen(kind, pcm))
        arrays[f"makethem_{kind}_head"], arrays[f"makethem_{kind}_tail"] = iq[:2048], iq[-2048:]
        manifest["makethem"].append({"kind": kind, "iq_bytes": int(iq.size), "iq_sha256": synth.digest(iq)})
    mods = {"ssb": (lambda: ref.ssbmod(True), LSB), "am": (ref.ammod, AM), "fm": (ref.fmmod, FM), "wbfm": (ref.wbfmmod, WBFM)}
    for name, (make, mode) in mods.items():
        iq = modulate(make(), pcm)
        arrays[f"mod_{name}_head"], arrays[f"mod_{name}_tail"] = iq[:2048], iq[-2048:]
        # the reference's OWN harness program of this modulator ({Am,Fm,WbFm,Ssb}Modulator/{am,fm,wbfm,ssb}.cc, built by
        # oracle/Makefile from its build script's file list): count.raw on stdin, int8 IQ on stdout -- the same bytes
        prog = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "prog_" + name)], stdin=open(SRC, "rb"), stdout=subprocess.PIPE, check=True).stdout
        assert synth.digest(np.frombuffer(prog, dtype=np.int8)) == synth.digest(iq), name
        manifest["modulator"].append({"kind": name, "iq_bytes": int(iq.size), "iq_sha256": synth.digest(iq),
                                      "program": {"ssb": "SsbModulator/ssb.cc", "am": "AmModulator/am.cc", "fm": "FmModulator/fm.cc",
                                                  "wbfm": "WbFmModulator/wbfm.cc"}[name],
                                      "program_sha256": synth.digest(np.frombuffer(prog, dtype=np.int8))})
        rx = ref.rx()
        rx.set_mode(mode)
        air = T.retune_minus_64k(iq)
        back = demodulate(rx, air)
        arrays[f"loop_{name}_pcm"] = back
        a, b = pcm.astype(np.float64), back.astype(np.float64)
        corr = max(abs(np.corrcoef(a[:39000], b[d:39000 + d])[0, 1]) for d in range(0, 400)) if b.std() > 0 else 0.0
        manifest["loop"].append({"kind": name, "mode": mode, "pcm_samples": int(back.size), "pcm_sha256": synth.digest(back),
                                 "air_sha256": synth.digest(air), "best_abs_correlation_with_count_raw": round(float(corr), 3)})
    np.savez_compressed(os.path.join(HERE, "golden_count.npz"), **arrays)
    with open(os.path.join(HERE, "golden_count.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print("wrote count.raw, golden_count.npz / .json:", {k: (len(v) if isinstance(v, list) else v) for k, v in manifest.items()})


if __name__ == "__main__":
    main()
