#!/usr/bin/env python3
"""Golden vectors from the reference's own smoke programs of the float filters: Filters/testFirFilter.cc and
Filters/testIirFilter.cc (compiled where they lie by oracle/Makefile, as buildTestFirFilter.sh / buildTestIirFilter.sh
do).  They run an impulse and a step of 19 samples through FirFilter {1,2,3,4,1,1,1,8}, IirFilter {1}/{0.5} and the
dc-removal IirFilter {1,-1}/{-0.95} -- the very filter the AM and SSB demodulators end in -- and print every output with
"%f"; the reference records no expected values, so they are produced here, by the reference itself.

Fixture: what the programs PRINTED (the six-decimal strings, per section), with the inputs and coefficients each section
ran on as numbers.  Build container only.

    python tests/golden/make_golden_filter_programs.py
"""
import json
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
IMPULSE = [1.0] + [0.0] * 18
STEP = [1.0] * 19
# section title as printed -> (numerator, denominator, input); the filter state is reset between sections
PROGRAMS = {
    "testFirFilter": [("Testing filter with impulse.", [1, 2, 3, 4, 1, 1, 1, 8], [], IMPULSE),
                      ("Testing filter with step.", [1, 2, 3, 4, 1, 1, 1, 8], [], STEP)],
    "testIirFilter": [("Testing filter with impulse.", [1], [0.5], IMPULSE),
                      ("Testing filter with step.", [1], [0.5], STEP),
                      ("Testing dc removal filter with impulse.", [1, -1], [-0.95], IMPULSE),
                      ("Testing dc removal filter with step.", [1, -1], [-0.95], STEP)],
}


def sections(text):
    out, cur = [], None
    for line in text.splitlines():
        if line.startswith("Testing"):
            cur = {"title": line.strip(), "printed": []}
            out.append(cur)
        else:
            m = re.match(r"sample\[(\d+)\] = (\S+)$", line)
            if m:
                assert int(m.group(1)) == len(cur["printed"])
                cur["printed"].append(m.group(2))
    return out


def main():
    man = {}
    for prog, plan in PROGRAMS.items():
        text = subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", prog)]).decode()
        secs = sections(text)
        assert [s["title"] for s in secs] == [p[0] for p in plan]
        for s, (_, b, a, x) in zip(secs, plan):
            assert len(s["printed"]) == len(x) == 19
            s.update({"numerator": b, "denominator": a, "input": x})
        man[prog] = secs
    with open(os.path.join(HERE, "golden_filter_programs.json"), "w") as f:
        json.dump(man, f, indent=1)
    print({k: [len(s["printed"]) for s in v] for k, v in man.items()})


if __name__ == "__main__":
    main()
