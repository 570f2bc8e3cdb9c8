"""The transmit side's PCM ring: BasebandDataProcessor's 16-slot ring with its
drop / repeat pacing.  Three implementations must agree block for block and counter for counter on
random schedules of writes, reads, starts and stops: the compiled reference (its class driven
without the reader thread), the oracle restatement, and the library's many-channel hrfd_txring
(host code: no GPU needed)."""
import ctypes as C

import numpy as np
import pytest

from hackrfdiags_amd import _lib


def _schedule(seed, n):
    rng = np.random.default_rng(seed)
    # sixteen writes first: the reference's ring slots are uninitialised memory until written
    # (the restatements zero them), so only then is every block a reader can get defined
    ops = ["w"] * 16
    for _ in range(n):
        x = rng.random()
        ops.append("w" if x < 0.47 else "r" if x < 0.94 else "start" if x < 0.97 else "stop")
    return ops, rng


def _drive(ring, ops, rng_seed):
    rng = np.random.default_rng(rng_seed)
    out = []
    for op in ops:
        if op == "w":
            ring.write(rng.integers(-32768, 32768, 512).astype(np.int16))
        elif op == "r":
            out.append(ring.read().copy())
        else:
            ring.set_running(op == "start")
    return np.array(out), ring.stats()


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_oracle_ring_equals_reference(oracle, ref, seed):
    ops, _ = _schedule(seed, 3000)
    a, sa = _drive(oracle.txring(), ops, 100 + seed)
    b, sb = _drive(ref.txring(), ops, 100 + seed)
    assert (a == b).all() and (sa == sb).all()
    assert sa[2] > 0 and sa[3] > 0                      # both the drop and the repeat branch ran


class _LibRing:
    """one channel of a 3-channel hrfd_txring (the other channels get a different schedule)"""

    def __init__(self, channel=1, n=3):
        self.L = _lib.load()
        self.h = C.c_void_p()
        _lib.check(self.L.hrfd_txring_create(n, C.byref(self.h)), "hrfd_txring_create")
        self.c, self.n = channel, n
        self.batch = np.zeros((n, 512), dtype=np.int16)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.hrfd_txring_destroy(self.h)

    def set_running(self, running):
        _lib.check(self.L.hrfd_txring_set_running(self.h, self.c, int(running)), "set_running")

    def write(self, pcm):
        _lib.check(self.L.hrfd_txring_write(self.h, self.c, pcm.ctypes.data_as(C.c_void_p)), "write")
        if pcm[0] & 1:                                   # unrelated traffic on another channel
            _lib.check(self.L.hrfd_txring_write(self.h, 0, pcm.ctypes.data_as(C.c_void_p)), "write")

    def read(self):
        _lib.check(self.L.hrfd_txring_read_batch(self.h, self.batch.ctypes.data_as(C.c_void_p)), "read_batch")
        return self.batch[self.c]

    def stats(self):
        out = np.zeros(6, dtype=np.uint32)
        _lib.check(self.L.hrfd_txring_stats(self.h, self.c, out.ctypes.data_as(C.POINTER(C.c_uint32))), "stats")
        return out


@pytest.mark.parametrize("seed", [5, 6])
def test_library_ring_equals_oracle(oracle, seed):
    ops, _ = _schedule(seed, 3000)
    a, sa = _drive(oracle.txring(), ops, 200 + seed)
    b, sb = _drive(_LibRing(), ops, 200 + seed)
    assert (a == b).all() and (sa == sb).all()
