#!/usr/bin/env python3
"""Golden vector from the reference's own test program of Decimator_int16: Filters/Int16/decimateAudio.cc (compiled where
it lies by oracle/Makefile, as its buildDecimateAudio.sh does) reads original32000.raw from its working directory, runs
every sample through Decimator_int16(80 taps, M = 4)::decimate and writes decimated8000.raw.  The reference commits the
input but no output for this int16 variant: the output is produced here, by the reference itself.

Fixture (small): the FIRST 40 000 samples of the reference's input file (the program reads what is there and runs on zeros
behind it), the prototype's 80 coefficients as numbers (read out of the program's table at generation time: data, not
source), the first 12 000 output samples and the digest of all 80 000.  Build container only.

    python tests/golden/make_golden_decimate_audio.py
"""
import json
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from hackrfdiags_amd import synth  # noqa: E402

REF = "/root/reference/radioDiags/Filters/Int16"
EXE = os.path.join(ROOT, "oracle", "_ref", "decimateAudio")
KEEP = 40000


def main():
    x = np.fromfile(os.path.join(REF, "original32000.raw"), dtype="<i2")[:KEEP]
    text = open(os.path.join(REF, "decimateAudio.cc")).read()
    body = text[text.index("h32000[]"):]
    body = body[body.index("{") + 1:body.index("}")]
    taps = np.array([float(t) for t in re.findall(r"-?\d+\.\d+(?:[eE][-+]?\d+)?", body)], dtype=np.float32)
    assert taps.size == 80
    with tempfile.TemporaryDirectory() as d:
        x.tofile(os.path.join(d, "original32000.raw"))
        subprocess.check_call([EXE], cwd=d)
        y = np.fromfile(os.path.join(d, "decimated8000.raw"), dtype="<i2")
    assert y.size == 80000
    np.savez_compressed(os.path.join(HERE, "golden_decimate_audio.npz"), input_head=x, taps=taps, output_head=y[:12000])
    man = {"program": "Filters/Int16/decimateAudio.cc (Decimator_int16, 80 taps, M = 4)", "input": "first 40000 samples of Filters/Int16/original32000.raw",
           "input_sha256": synth.digest(x), "output_samples": int(y.size), "output_sha256": synth.digest(y),
           "output_nonzero_until": int(np.nonzero(y)[0][-1]) if y.any() else 0}
    with open(os.path.join(HERE, "golden_decimate_audio.json"), "w") as f:
        json.dump(man, f, indent=1)
    print(man)


if __name__ == "__main__":
    main()
