"""CPU-only sanitizer jobs for the host-only pieces of libhrfd (round 5; GPU AddressSanitizer is not available on the
pool, and the device code has its own stress build).  The transmit ring (hackrfdiags_amd/csrc/hrfd_txring.hip: plain C++
behind the C ABI, two threads by design -- the PCM reader and the transmit callback, BasebandDataProcessor.cc:476-606, 869)
is compiled as it stands into tests/cpp/san_txring.cc under -fsanitize=address,undefined and under -fsanitize=thread
and run: writer against reader over three channels, stop and restart, the error returns."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "cpp", "san_txring.cc")


@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_txring_under_sanitizers(tmp_path, san):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "san_txring")
    cmd = ["g++", "-x", "c++", "-std=c++17", "-g", "-O1", f"-fsanitize={san}", "-fno-sanitize-recover=all", "-o", exe, SRC, "-lpthread"]
    b = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.join(HERE, "cpp"))
    if b.returncode != 0 and ("cannot find" in b.stderr or "unrecognized" in b.stderr):
        pytest.skip("this toolchain has no runtime for -fsanitize=" + san)
    assert b.returncode == 0, b.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1", TSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "san_txring ok" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
