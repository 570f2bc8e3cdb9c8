"""The drop-in boundary against the reference's OWN translation units (SURVEY 8b "who calls it": unchanged callers,
only the libraries are swapped).

The reference application is built by one g++ command over its .cc files (radioDiags/buildRadioDiags.sh:50) and
hdr_diags/Radio.h:16,18,28 include "IqDataProcessor.h", "DataProvider.h", "BasebandDataProcessor.h" by quoted names from
their own directory: no -I order replaces those three.  The shim classes are therefore LAYOUT-CONTAINED in the
reference's (shim/hrfd_shim_layout.h), and these tests pin that with the reference's own files:

  CPU, where /root/reference exists (the build container; skipped elsewhere):
    * the sizes in hrfd_shim_layout.h are the reference's, recomputed from its headers;
    * Radio.cc, diagUi.cc, DataConsumer.cc, AutomaticGainControl.cc, FrequencyScanner.cc, FrequencySweeper.cc, radioApp.cc,
      console.cc, ... compile unchanged both ways: beside the reference's three headers (the mixed build) and with those three
      hidden (every class declaration the shim's);
    * the whole application -- buildRadioDiags.sh's file list minus the files the shim replaces -- links against
      hrfd_shim.cc + libhrfd.so with NOTHING unresolved but libhackrf's own entry points (libhackrf needs libusb, which
      this image lacks: it is the one thing not linked);
  GPU (oracle/_ref/dropin_app, built here by oracle/Makefile and shipped prebuilt like the rest of oracle/_ref):
    * the reference's unchanged DataConsumer.cc + MessageQueue.cc + UdpClient.cc and the Radio constructor's wiring,
      compiled against the reference's headers, run on the shim: blocks through DataConsumer::acceptData -> consumer
      thread -> IqDataProcessor::acceptIqData -> PCM callback equal the oracle's PCM in all five modes; the transmit
      side through the reference-declared BasebandDataProcessor (its own stdin reader thread) and DataProvider."""
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

from hackrfdiags_amd import synth
from tests.reflib import AM, FM, WBFM, LSB, USB, ORACLE_DIR
from tests import toolsupport

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "hackrfdiags_amd", "csrc", "shim")
REF = "/root/reference/radioDiags"
APP = os.path.join(ORACLE_DIR, "_ref", "dropin_app")
BLOCK = 262144

needs_reference = pytest.mark.skipif(not os.path.isdir(REF), reason="/root/reference absent (GPU box)")

# buildRadioDiags.sh:12-28 (CcFiles) minus what the shim replaces (IqDataProcessor.cc, BasebandDataProcessor.cc,
# DataProvider.cc and the squelch chain that only IqDataProcessor.cc used: Squelch.cc, SignalDetector.cc,
# SignalTracker.cc, Decimator_int16.cc).  DbfsCalculator.cc stays: the AGC uses it.
APP_UNITS = ["radioApp.cc", "DataConsumer.cc", "FrequencyScanner.cc", "FrequencySweeper.cc", "DbfsCalculator.cc",
             "AutomaticGainControl.cc", "Radio.cc", "console.cc", "diagUi.cc", "UdpClient.cc", "MessageQueue.cc"]
# buildRadioDiags.sh:30-43
REF_INC_DIRS = ["hackRf", "hdr_diags", "Filters", "Filters/Int16", "Nco", "AmDemodulator", "FmDemodulator",
                "WbFmDemodulator", "SsbDemodulator", "AmModulator", "FmModulator", "WbFmModulator", "SsbModulator"]
REPLACED = ["IqDataProcessor", "BasebandDataProcessor", "DataProvider", "AmDemodulator", "FmDemodulator",
            "WbFmDemodulator", "SsbDemodulator", "AmModulator", "FmModulator", "WbFmModulator", "SsbModulator", "Nco"]


def _ref_includes():
    out = []
    for d in REF_INC_DIRS:
        out += ["-I", os.path.join(REF, d)]
    return out


def _layout_constants():
    text = open(os.path.join(SHIM, "hrfd_shim_layout.h")).read()
    return {m.group(1): int(m.group(2)) for m in re.finditer(r"#define HRFD_REF_SIZEOF_(\w+)\s+(\d+)", text)}


@needs_reference
def test_layout_header_holds_the_reference_sizes(tmp_path):
    """hrfd_shim_layout.h's numbers, recomputed from the reference's headers as they lie"""
    probe = tmp_path / "probe.cc"
    lines = ['#include <stdio.h>', '#include "IqDataProcessor.h"', '#include "BasebandDataProcessor.h"',
             '#include "DataProvider.h"', '#include "Nco.h"', 'int main(){']
    lines += [f'printf("{c} %zu %zu %d\\n", sizeof({c}), alignof({c}), (int)__is_polymorphic({c}));' for c in REPLACED]
    lines += ['return 0;}']
    probe.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.check_call(["g++", "-w", "-o", str(exe), str(probe)] + _ref_includes())
    got = {}
    for ln in subprocess.check_output([str(exe)], text=True).splitlines():
        name, size, align, poly = ln.split()
        got[name] = int(size)
        assert int(align) <= 8 and int(poly) == 0
    assert got == _layout_constants()


def test_shim_classes_fit_inside_the_reference_classes(tmp_path):
    """the static_asserts of hrfd_shim.cc, and the same comparison at run time (no reference needed: the constants)"""
    probe = tmp_path / "probe.cc"
    lines = ['#include <stdio.h>', '#include "IqDataProcessor.h"', '#include "BasebandDataProcessor.h"',
             '#include "DataProvider.h"', '#include "Nco.h"', '#include "hrfd_shim_layout.h"', 'int main(){']
    lines += [f'printf("{c} %zu %d\\n", sizeof({c}), HRFD_REF_SIZEOF_{c});' for c in REPLACED]
    lines += ['return 0;}']
    probe.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.check_call(["g++", "-w", "-o", str(exe), str(probe), "-I", SHIM, "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "tests", "cpp")])
    for ln in subprocess.check_output([str(exe)], text=True).splitlines():
        name, mine, theirs = ln.split()
        assert int(mine) <= int(theirs), f"{name}: shim {mine} B > reference {theirs} B"


def _hidden_header_dir(tmp_path):
    """hdr_diags with the three replaced headers hidden: a directory of symbolic links (nothing is copied)"""
    d = tmp_path / "hdr_app"
    d.mkdir()
    for f in os.listdir(os.path.join(REF, "hdr_diags")):
        if f not in ("IqDataProcessor.h", "BasebandDataProcessor.h", "DataProvider.h"):
            os.symlink(os.path.join(REF, "hdr_diags", f), d / f)
    return str(d)


@needs_reference
@pytest.mark.parametrize("flavour", ["mixed", "shim_only"])
def test_reference_application_units_compile_unchanged(tmp_path, flavour):
    """`mixed`: the reference's include line with the shim in front -- Radio.h still finds the reference's three
    headers beside itself; `shim_only`: those three hidden, every replaced class is the shim's."""
    if flavour == "mixed":
        inc = ["-I", SHIM, "-I", os.path.join(ROOT, "include")] + _ref_includes()
    else:
        inc = ["-I", SHIM, "-I", os.path.join(ROOT, "include"), "-I", os.path.join(REF, "hackRf"),
               "-I", _hidden_header_dir(tmp_path)]
    for unit in APP_UNITS:
        src = os.path.join(REF, "src_diags", unit)
        r = subprocess.run(["g++", "-O3", "-w", "-fsyntax-only", src] + inc, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0, f"{unit} ({flavour}):\n{r.stderr[-2000:]}"


@needs_reference
@pytest.mark.parametrize("line", ["reference_headers_only", "documented_build_line"])
def test_whole_reference_application_links_against_the_shim(tmp_path, line):
    """Every object of the application, compiled from the reference's unchanged sources -- against the reference's own
    include line (every replaced class declared by the REFERENCE's header: the most adverse mix), and against
    INTEGRATION.md's line (the shim in front: the (de)modulators are the shim's, the three hdr_diags classes the
    reference's) -- plus hrfd_shim.cc, -lhrfd, -lamdhip64: the only unresolved symbols are libhackrf's."""
    front = [] if line == "reference_headers_only" else ["-I", SHIM, "-I", os.path.join(ROOT, "include")]
    objs = []
    for unit in APP_UNITS:
        o = str(tmp_path / (unit[:-3] + ".o"))
        subprocess.check_call(["g++", "-O3", "-w", "-c", "-o", o, os.path.join(REF, "src_diags", unit)] + front +
                              _ref_includes())
        objs.append(o)
    shim_o = str(tmp_path / "hrfd_shim.o")
    subprocess.check_call(["g++", "-O3", "-std=c++17", "-Wall", "-c", "-o", shim_o, os.path.join(SHIM, "hrfd_shim.cc"),
                           "-I", os.path.join(ROOT, "include"), "-I", SHIM, "-I", os.path.join(REF, "hdr_diags")])
    lib = os.path.join(ROOT, "hackrfdiags_amd", "lib")
    r = subprocess.run(["g++", "-O3", "-o", str(tmp_path / "radioDiags")] + objs +
                       [shim_o, "-L", lib, "-lhrfd", "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-lpthread", "-lrt",
                        "-Wl,--no-demangle"], stderr=subprocess.PIPE, text=True)
    undefined = set(re.findall(r"undefined reference to `([^']+)'", r.stderr))
    assert undefined, "libhackrf is not linked, so its entry points must be what is missing"
    not_hackrf = sorted(s for s in undefined if not s.startswith("hackrf_"))
    assert not not_hackrf, f"unresolved besides libhackrf: {not_hackrf[:20]}"
    # and nothing is defined twice (the reference's implementation files of the replaced classes are not in the list)
    assert "multiple definition" not in r.stderr
    # the application objects do reference the replaced classes (the test would be empty otherwise)
    syms = subprocess.check_output(["nm", "-u", "-C", objs[APP_UNITS.index("Radio.cc")]], text=True)
    for cls in ("IqDataProcessor::IqDataProcessor", "BasebandDataProcessor::BasebandDataProcessor",
                "DataProvider::DataProvider", "WbFmDemodulator::WbFmDemodulator", "SsbModulator::SsbModulator"):
        assert cls in syms
    defined = subprocess.check_output(["nm", "--defined-only", "-C", shim_o], text=True)
    for cls in ("IqDataProcessor::acceptIqData", "BasebandDataProcessor::getIqData", "DataProvider::getIqData"):
        assert cls in defined


@needs_reference
def test_dropin_app_sees_the_reference_headers():
    subprocess.check_call(["make", "-C", ORACLE_DIR, "dropin"], stdout=subprocess.DEVNULL)
    err = subprocess.run([APP, "sizes"], stderr=subprocess.PIPE, text=True, check=True).stderr
    seen = {m.group(1): int(m.group(2)) for m in re.finditer(r"sizeof (\w+) (\d+)", err)}
    want = _layout_constants()
    assert seen == {k: v for k, v in want.items() if k in seen} and len(seen) == 11


def _need_app():
    if not os.path.exists(APP):
        pytest.skip("oracle/_ref/dropin_app not built (needs /root/reference at build time)")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [AM, FM, WBFM, LSB, USB])
def test_reference_data_consumer_drives_the_shim(oracle, mode):
    """DataConsumer::acceptData (DataConsumer.cc:219-262) -> message queue -> dataConsumerProcedure (:319-351) ->
    IqDataProcessor::acceptIqData of the SHIM on an object allocated with the REFERENCE's sizeof -> PCM callback ->
    stdout (radioApp.cc:103-111).  Eight blocks; PCM = the oracle's, bit for bit."""
    _need_app()
    n = 8
    x = synth.make_input("fmtone", 11 + mode, n)
    r = subprocess.run([APP, "rx", str(mode), str(n)], input=x.tobytes(), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    got = np.frombuffer(r.stdout, dtype=np.int16)
    o = oracle.rx(); o.set_mode(mode)
    want = np.concatenate([o.process(x[b * BLOCK:(b + 1) * BLOCK])[0] for b in range(n)])
    assert len(got) == len(want) == n * 512 and (got == want).all()
    assert f"{n} magnitude callbacks".encode() in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [AM, FM, WBFM, LSB, USB])
def test_reference_data_consumer_hands_short_transfers_to_the_shim(oracle, mode):
    """USB transfers that end early (hackRf/hackrf.c:1443: valid_length = actual_length) through the reference's
    unchanged DataConsumer::acceptData -- which counts them and passes them on (DataConsumer.cc:229-241) and clips one
    that is too long -- into the shim's IqDataProcessor::acceptIqData: no abort, the oracle's PCM call for call."""
    _need_app()
    lengths = [262144, 261632, 262144, 16896, 245248, 1000, 262144, 262144 + 1024, 512, 262144]
    x = synth.make_input("fmtone", 40 + mode, 8)
    assert sum(lengths) <= len(x)
    r = subprocess.run([APP, "rx", str(mode), str(len(lengths)), ",".join(str(n) for n in lengths)],
                       input=x[:sum(lengths)].tobytes(), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-2000:]
    got = np.frombuffer(r.stdout, dtype=np.int16)
    o = oracle.rx(); o.set_mode(mode)
    want, off = [], 0
    for n in lengths:
        want.append(o.process(x[off:off + min(n, BLOCK)])[0])   # (a transfer longer than the buffer is clipped, the rest is lost)
        off += n
    want = np.concatenate(want)
    assert len(got) == len(want) and (got == want).all()
    assert f"{len(lengths)} magnitude callbacks".encode() in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4, 5])
def test_reference_declared_baseband_processor_on_the_shim(oracle, mode):
    """BasebandDataProcessor allocated through the reference's header (17 664 B), start() (the reader thread takes
    stdin), getIqData x 6 (Radio.cc:3221-3227): = the oracle's ring model feeding the oracle's modulator."""
    _need_app()
    n_w, n_r = 8, 6
    pcm = synth.lcg_pcm(31 + mode, n_w * 512)
    r = subprocess.run([APP, "tx", str(mode), str(n_w), str(n_r)], input=pcm.tobytes(), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    got = np.frombuffer(r.stdout, dtype=np.int8).reshape(n_r, BLOCK)
    ring = oracle.txring()
    ring.set_running(True)
    for w in range(n_w):
        ring.write(pcm[w * 512:(w + 1) * 512])
    mod = {1: lambda: oracle.ammod(), 2: lambda: oracle.fmmod(), 3: lambda: oracle.wbfmmod(),
           4: lambda: oracle.ssbmod(True), 5: lambda: oracle.ssbmod(False)}.get(mode, lambda: None)()
    want = []
    for _ in range(n_r):
        blk = ring.read()
        want.append(np.full(BLOCK, 64, dtype=np.int8) if mod is None else mod.process(blk))
    want = np.stack(want)
    assert (got == want).all()                      # every mode; FM: libm cosf/sinf in the reference, restated on the device (round 5)


@pytest.mark.gpu
def test_reference_declared_data_provider_on_the_shim(tmp_path):
    """DataProvider allocated through the reference's header (272 B): `load iqfile` + cyclic getIqData
    (DataProvider.cc:174-231) = the index arithmetic tests/test_tools.py pins to the compiled DataProvider"""
    _need_app()
    image = synth.lcg_bytes(77, 100003).view(np.int8)
    path = tmp_path / "x.iq"
    path.write_bytes(image.tobytes())
    n, nbytes = 5, 65536
    r = subprocess.run([APP, "file", str(path), str(n), str(nbytes)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    got = np.frombuffer(r.stdout, dtype=np.int8).reshape(n, nbytes)
    idx = 0
    for k in range(n):
        want, idx = toolsupport.playback_model(image, idx, nbytes)
        assert (got[k] == want).all()
