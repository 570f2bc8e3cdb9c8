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


SHIM = os.path.join(os.path.dirname(HERE), "hackrfdiags_amd", "csrc", "shim")


def test_shim_against_a_mock_abi_under_sanitizers(tmp_path):
    """The drop-in classes (hackrfdiags_amd/csrc/shim/hrfd_shim.cc) and the miniature of the reference's wiring
    (tests/cpp/shim_demo.cc) compiled against a MOCK of the C ABI (tests/cpp/mock_hrfd.cc: every output buffer written in
    full, every input byte read, nothing real computed; the transmit ring is the real one) under
    -fsanitize=address,undefined -- the shim's own host logic without a GPU: the heap block behind the layout-contained
    IqDataProcessor, the callbacks' buffers, BasebandDataProcessor's ring schedule, file playback, the Fs/4 helpers."""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "shim_demo_san")
    root = os.path.dirname(HERE)
    cmd = ["g++", "-std=c++17", "-g", "-O1", "-x", "c++", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe,
           os.path.join(HERE, "cpp", "shim_demo.cc"), os.path.join(SHIM, "hrfd_shim.cc"), os.path.join(HERE, "cpp", "mock_hrfd.cc"),
           "-I", os.path.join(root, "include"), "-I", SHIM, "-I", os.path.join(HERE, "cpp"), "-lpthread"]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0 and ("cannot find" in b.stderr or "unrecognized" in b.stderr):
        pytest.skip("this toolchain has no sanitizer runtime")
    assert b.returncode == 0, b.stderr[-3000:]
    iq = bytes((i * 37 + 11) & 0xFF for i in range(1 << 20))
    pcm = bytes((i * 13 + 5) & 0xFF for i in range(24 * 1024))
    f = tmp_path / "play.iq"
    f.write_bytes(pcm)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1")
    runs = [(["3", "outer", "262144"], iq, 4 * 512 * 2), (["1", "outer", "262144"], iq, 4 * 512 * 2), (["5", "outer", "65536"], iq, 16 * 128 * 2),
            (["2", "inner", "32768"], iq, 32 * 512 * 2), (["4", "inner", "32768"], iq, 32 * 512 * 2),
            (["4", "ssbmod", "0"], pcm, 24 * 262144), (["500", "ammod", "0"], pcm, 24 * 262144), (["1200", "fmmod", "0"], pcm, 24 * 262144),
            (["30000", "wbfmmod", "0"], pcm, 24 * 262144), (["3", "bbp", "0", "rwwwwwwwwwwwwwwwwsrrrwrwrwrrrrwwwwprr"], pcm, 12 * 262144),
            (["1", "fs4", "4096"], iq, 4096), (["0", "fs4", "4096"], iq, 4096), (["5", "provider", "3000", str(f)], b"", 15000)]
    for args, data, n_out in runs:
        r = subprocess.run([exe] + args, input=data, capture_output=True, env=env, timeout=300)
        assert r.returncode == 0, (args, r.stderr[-3000:])
        assert len(r.stdout) == n_out, (args, len(r.stdout), n_out)
