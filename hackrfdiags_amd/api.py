"""Python mirror of the C ABI (include/hrfd.h) -- thin ctypes plumbing used by
the tests and bench.py.  Names follow the reference: an Rx is C channels of
IqDataProcessor + demodulators; process_block == acceptIqData."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import HrfdError, check  # noqa: F401

NONE, AM, FM, WBFM, LSB, USB = range(6)
ALL = 0xFFFFFFFF
BLOCK_BYTES = 262144


def pcm_capacity(block_bytes: int) -> int:
    """row length of the PCM output for a block length: ceil(block_bytes / 512) (hrfd_rx_pcm_capacity)"""
    return (int(block_bytes) + 511) // 512


def iq256_capacity(block_bytes: int) -> int:
    """row length in bytes of the 256 kS/s dump for a block length: 2 * ceil(block_bytes / 16) (hrfd_rx_iq256_capacity)"""
    return 2 * ((int(block_bytes) // 2 + 7) // 8)


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return C.c_void_p(a.ctypes.data)
    return C.c_void_p(int(a))          # raw device pointer (e.g. torch tensor .data_ptr())


class Rx:
    """n_channels receive chains (hrfd_rx_*)."""

    def __init__(self, n_channels: int, device: int = -1):
        self.L = _lib.load()
        self.n = int(n_channels)
        h = C.c_void_p()
        check(self.L.hrfd_rx_create(self.n, device, C.byref(h)), "hrfd_rx_create")
        self.h = h
        self.gain_db = 0

    def close(self):
        if getattr(self, "h", None):
            self.L.hrfd_rx_destroy(self.h)
            self.h = None

    __del__ = close

    def set_mode(self, mode, channel=ALL):
        check(self.L.hrfd_rx_set_mode(self.h, channel, mode), "hrfd_rx_set_mode")

    def set_gain(self, mode, gain, channel=ALL):
        check(self.L.hrfd_rx_set_gain(self.h, channel, mode, C.c_float(gain)), "hrfd_rx_set_gain")

    def set_threshold(self, threshold, channel=ALL):
        check(self.L.hrfd_rx_set_threshold(self.h, channel, threshold), "hrfd_rx_set_threshold")

    def reset_demod(self, mode, channel=ALL):
        check(self.L.hrfd_rx_reset_demod(self.h, channel, mode), "hrfd_rx_reset_demod")

    def process_block(self, iq: np.ndarray, n_blocks: int = 1, want_iq256: bool = False):
        """iq: int8 [C, n_blocks, block_bytes] (or flat).  Host buffers in, host arrays out:
        (pcm [C,B,npcm], n_pcm [C,B], magnitude [C,B], allowed [C,B], iq256 [C,B,bb/8] | None)"""
        iq = np.ascontiguousarray(iq, dtype=np.int8).reshape(self.n, n_blocks, -1)
        bb = iq.shape[2]
        npcm = pcm_capacity(bb)                            # rows: what a call of any even length can complete at most
        pcm = np.zeros((self.n, n_blocks, npcm), dtype=np.int16)
        n_pcm = np.zeros((self.n, n_blocks), dtype=np.uint32)
        mag = np.zeros((self.n, n_blocks), dtype=np.uint32)
        allowed = np.zeros((self.n, n_blocks), dtype=np.uint8)
        iq256 = np.zeros((self.n, n_blocks, iq256_capacity(bb)), dtype=np.int8) if want_iq256 else None
        check(self.L.hrfd_rx_process_block(self.h, _ptr(iq), bb, n_blocks, self.gain_db, _ptr(pcm),
                                           _ptr(n_pcm), _ptr(mag), _ptr(allowed), _ptr(iq256)),
              "hrfd_rx_process_block")
        return pcm, n_pcm, mag, allowed, iq256

    def reduce_sample_rate(self, iq: np.ndarray) -> np.ndarray:
        """IqDataProcessor::reduceSampleRate for one block of every channel: iq int8 [C, block_bytes] -> the 256 kS/s
        stream int8 [C, block_bytes / 8] (with the Fs/4 rotation); only the decimator pipelines advance"""
        iq = np.ascontiguousarray(iq, dtype=np.int8).reshape(self.n, -1)
        out = np.zeros((self.n, iq256_capacity(iq.shape[1])), dtype=np.int8)
        check(self.L.hrfd_rx_reduce_sample_rate(self.h, _ptr(iq), iq.shape[1], _ptr(out)), "hrfd_rx_reduce_sample_rate")
        return out

    def process_device(self, d_iq, channel_stride, block_bytes, n_blocks, d_pcm, d_n_pcm=None,
                       d_magnitude=None, d_allowed=None, d_iq256=None, stream=None):
        """All pointers are device addresses (ints); asynchronous."""
        check(self.L.hrfd_rx_process_device(self.h, _ptr(d_iq), channel_stride, block_bytes, n_blocks,
                                            self.gain_db, _ptr(d_pcm), _ptr(d_n_pcm), _ptr(d_magnitude),
                                            _ptr(d_allowed), _ptr(d_iq256), _ptr(stream)),
              "hrfd_rx_process_device")

    def pending_samples(self) -> int:
        """IQ samples the front end holds back after the calls so far (0..7): the next call of bb bytes completes
        (pending + bb // 2) // 8 samples at 256 kS/s (IqDataProcessor::reduceSampleRate's count)"""
        v = C.c_uint32(0)
        check(self.L.hrfd_rx_pending_samples(self.h, C.byref(v)), "hrfd_rx_pending_samples")
        return int(v.value)

    def sync(self) -> int:
        """waits for the last process_device; returns the number of channels that did not commit"""
        v = C.c_uint32(0)
        check(self.L.hrfd_rx_sync(self.h, C.byref(v)), "hrfd_rx_sync")
        return int(v.value)

    def failed_channels(self) -> np.ndarray:
        """uint8 [n_channels]: != 0 where the channel did not commit in the launch sync() last waited for"""
        out = np.zeros(self.n, dtype=np.uint8)
        check(self.L.hrfd_rx_failed_channels(self.h, out.ctypes.data, self.n), "hrfd_rx_failed_channels")
        return out

    # test hooks
    def debug_set_atan(self, mode: int):
        """-1 automatic, 0 force the atan2 table gather, 1 require the arithmetic atan2 kernel."""
        check(self.L.hrfd_rx_debug_set_atan(self.h, int(mode)), "hrfd_rx_debug_set_atan")

    def debug_atan_eval(self, tab: bool = False):
        """The arithmetic atan2 of the WBFM kernels over all (q, i): float32 [256][256]; tab: the
        first-octant-table variant of k_rx_wbfm_flow instead of the polynomial one."""
        out = np.zeros((256, 256), dtype=np.float32)
        fn = (self.L.hrfd_rx_debug_atan_eval_quad if tab == "quad" else
              self.L.hrfd_rx_debug_atan_eval_tab if tab else self.L.hrfd_rx_debug_atan_eval)
        check(fn(self.h, out.ctypes.data_as(C.POINTER(C.c_float))), "hrfd_rx_debug_atan_eval")
        return out

    def debug_ragged(self):
        """(the handle left the 512-byte grid, launches that ran on the general-length kernel k_rx_ragged)"""
        off, n = C.c_int32(0), C.c_ulonglong(0)
        check(self.L.hrfd_rx_debug_ragged(self.h, C.byref(off), C.byref(n)), "hrfd_rx_debug_ragged")
        return bool(off.value), int(n.value)

    def debug_set_run_len(self, blocks: int):
        """consecutive blocks of a channel per WBFM workgroup (0 = automatic)"""
        check(self.L.hrfd_rx_debug_set_run_len(self.h, int(blocks)), "hrfd_rx_debug_set_run_len")

    def debug_set_warm(self, warm: int):
        check(self.L.hrfd_rx_debug_set_warm(self.h, warm), "hrfd_rx_debug_set_warm")

    def debug_enable_timing(self, slots=1):
        """HIP events around the demodulator kernels of every launch (slot = launch % slots)."""
        check(self.L.hrfd_rx_debug_enable_timing(self.h, int(slots)), "hrfd_rx_debug_enable_timing")

    def debug_timing_every(self, n: int):
        """bracket only every n-th launch with events (the others run back to back, as in a host that does not measure)"""
        check(self.L.hrfd_rx_debug_timing_every(self.h, int(n)), "hrfd_rx_debug_timing_every")

    def debug_kernel_ms(self, slot=0) -> float:
        ms = C.c_float(0)
        check(self.L.hrfd_rx_debug_kernel_ms(self.h, int(slot), C.byref(ms)), "hrfd_rx_debug_kernel_ms")
        return float(ms.value)

    def debug_stamps(self, groups: int, read: bool = False):
        """groups > 0, read=False: allocate stamp storage; read=True: fetch [groups, 48] uint64 (kDbgSlots)."""
        if not read:
            check(self.L.hrfd_rx_debug_stamps(self.h, groups, None), "hrfd_rx_debug_stamps")
            return None
        out = np.zeros((groups, 48), dtype=np.uint64)
        check(self.L.hrfd_rx_debug_stamps(self.h, groups, _ptr(out)), "hrfd_rx_debug_stamps")
        return out

    def debug_set_stream(self, kernel):
        """test hook: WBFM batches on 0 / False = the block kernel k_rx_wbfm, anything else = k_rx_wbfm_flow where it
        applies (the default)"""
        check(self.L.hrfd_rx_debug_set_stream(self.h, int(kernel)), "hrfd_rx_debug_set_stream")

    def debug_set_fir_flow(self, mode: int):
        """test hook: AM / SSB / FM batches on the flow kernel's FIR modes: -1 automatic, 0 never, 1 always, 2 always and one
        launch per kind (a bank of several kinds does not take k_rx_flow_bank)"""
        check(self.L.hrfd_rx_debug_set_fir_flow(self.h, int(mode)), "hrfd_rx_debug_set_fir_flow")

    def debug_set_gated(self, on: bool):
        """test hook: False = no gated second pass on the device (closed gates in a batch go back to the host's replay)"""
        check(self.L.hrfd_rx_debug_set_gated(self.h, int(bool(on))), "hrfd_rx_debug_set_gated")

    def debug_expire(self, where: int):
        """test hook: workgroup 0 of the next k_rx_wbfm_flow launch treats its wait `where` (1..6) as expired"""
        check(self.L.hrfd_rx_debug_expire(self.h, int(where)), "hrfd_rx_debug_expire")

    def debug_set_stagger(self, units: int):
        check(self.L.hrfd_rx_debug_set_stagger(self.h, units), "hrfd_rx_debug_set_stagger")

    def debug_counters(self):
        """[repairs, gate_viol, spec_viol, committed, total_repairs, total_uncommitted_launches,
        total_launches, host_replays] -- the first four describe the last launch."""
        out = (C.c_uint32 * 8)()
        check(self.L.hrfd_rx_debug_counters(self.h, out), "hrfd_rx_debug_counters")
        return list(out)


class SingleChannelRx:
    """One channel with the call shape the golden checks use (mirrors one
    IqDataProcessor::acceptIqData call per process())."""

    def __init__(self, device: int = -1):
        self.rx = Rx(1, device)

    @property
    def gain_db(self):
        return self.rx.gain_db

    @gain_db.setter
    def gain_db(self, v):
        self.rx.gain_db = int(v)

    def set_mode(self, mode):
        self.rx.set_mode(mode)

    def set_gain(self, mode, gain):
        self.rx.set_gain(mode, gain)

    def set_threshold(self, t):
        self.rx.set_threshold(t)

    def process(self, iq):
        pending = self.rx.pending_samples()
        pcm, n_pcm, mag, allowed, iq256 = self.rx.process_block(iq, 1, want_iq256=True)
        n = int(n_pcm[0, 0])
        n256 = 2 * ((pending + np.asarray(iq).size // 2) // 8)    # decimatedByteCount of this call
        return pcm[0, 0, :n].copy(), int(mag[0, 0]), bool(allowed[0, 0]), iq256[0, 0, :n256]


class Demod:
    """n_channels instances of one demodulator class on 256 kS/s mixed IQ
    (hrfd_demod_*; mirrors X::acceptIqData / setDemodulatorGain / resetDemodulator)."""

    def __init__(self, mode: int, n_channels: int = 1, device: int = -1):
        self.L = _lib.load()
        self.n = int(n_channels)
        h = C.c_void_p()
        check(self.L.hrfd_demod_create(mode, self.n, device, C.byref(h)), "hrfd_demod_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.hrfd_demod_destroy(self.h)
            self.h = None

    __del__ = close

    def reset(self, channel=ALL):
        check(self.L.hrfd_demod_reset(self.h, channel), "hrfd_demod_reset")

    def set_gain(self, gain, channel=ALL):
        check(self.L.hrfd_demod_set_gain(self.h, channel, C.c_float(gain)), "hrfd_demod_set_gain")

    def set_sideband(self, lsb, channel=ALL):
        check(self.L.hrfd_demod_set_sideband(self.h, channel, int(bool(lsb))), "hrfd_demod_set_sideband")

    def process(self, iq256):
        """iq256: int8 [C, bytes] (or flat for C == 1) -> pcm int16 [C, bytes/64]"""
        iq256 = np.ascontiguousarray(iq256, dtype=np.int8).reshape(self.n, -1)
        nb = iq256.shape[1]
        pcm = np.zeros((self.n, (nb + 63) // 64), dtype=np.int16)
        n_pcm = np.zeros(self.n, dtype=np.uint32)
        check(self.L.hrfd_demod_process(self.h, _ptr(iq256), nb, _ptr(pcm), _ptr(n_pcm)), "hrfd_demod_process")
        assert (n_pcm == n_pcm[0]).all()                   # the channels of a handle have seen the same lengths
        pcm = pcm[:, :int(n_pcm[0])]
        return pcm if self.n > 1 else pcm[0]


MOD_SSB, MOD_INTERP, MOD_AM, MOD_FM, MOD_WBFM = 1, 2, 3, 4, 5
MOD_SIG_AM, MOD_SIG_DSB, MOD_SIG_PM, MOD_SIG_FM = 6, 7, 8, 9   # signals/{am,dsb,pm,fm}.cc | interpolateSignal


class Ingest:
    """Pinned-memory, double-buffered block transport in front of an Rx (hrfd_ingest_*)."""

    def __init__(self, rx: "Rx", block_bytes: int, n_blocks: int, n_slots: int = 2):
        self.L = _lib.load()
        self.rx = rx
        self.block_bytes, self.n_blocks, self.C = int(block_bytes), int(n_blocks), rx.n
        h = C.c_void_p()
        check(self.L.hrfd_ingest_create(rx.h, self.block_bytes, self.n_blocks, int(n_slots), C.byref(h)),
              "hrfd_ingest_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.hrfd_ingest_destroy(self.h)
            self.h = None

    __del__ = close

    def acquire(self) -> np.ndarray:
        """the next free slot's pinned input buffer as int8 [C, n_blocks, block_bytes] (a view)"""
        p = C.c_void_p()
        check(self.L.hrfd_ingest_acquire(self.h, C.byref(p)), "hrfd_ingest_acquire")
        n = self.C * self.n_blocks * self.block_bytes
        buf = (C.c_int8 * n).from_address(p.value)
        return np.frombuffer(buf, dtype=np.int8).reshape(self.C, self.n_blocks, self.block_bytes)

    def submit(self, gain_db: int = 0):
        check(self.L.hrfd_ingest_submit(self.h, int(gain_db)), "hrfd_ingest_submit")

    def collect(self):
        """(pcm [C, B, bytes/512], n_pcm [C, B], magnitude [C, B], allowed [C, B]) -- copies"""
        ps = [C.c_void_p() for _ in range(4)]
        check(self.L.hrfd_ingest_collect(self.h, *[C.byref(p) for p in ps]), "hrfd_ingest_collect")
        units = self.C * self.n_blocks
        npcm = pcm_capacity(self.block_bytes)

        def view(p, ctype, dtype, count):
            return np.frombuffer((ctype * count).from_address(p.value), dtype=dtype).copy()

        pcm = view(ps[0], C.c_int16, np.int16, units * npcm).reshape(self.C, self.n_blocks, npcm)
        n_pcm = view(ps[1], C.c_uint32, np.uint32, units).reshape(self.C, self.n_blocks)
        mag = view(ps[2], C.c_uint32, np.uint32, units).reshape(self.C, self.n_blocks)
        allowed = view(ps[3], C.c_uint8, np.uint8, units).reshape(self.C, self.n_blocks)
        return pcm, n_pcm, mag, allowed

    def replayed(self) -> int:
        n = C.c_uint64(0)
        check(self.L.hrfd_ingest_replayed(self.h, C.byref(n)), "hrfd_ingest_replayed")
        return int(n.value)


def fanout_channel_range(n_channels: int, n_shards: int, shard: int):
    """(first, count) of a shard: hrfd_fanout_channel_range (pure arithmetic, needs no GPU)"""
    L = _lib.load()
    first, count = C.c_uint32(0), C.c_uint32(0)
    check(L.hrfd_fanout_channel_range(int(n_channels), int(n_shards), int(shard), C.byref(first), C.byref(count)),
          "hrfd_fanout_channel_range")
    return int(first.value), int(count.value)


class Fanout:
    """One process, several devices: n_channels receive chains sharded over `devices` (hrfd_fanout_*)."""

    def __init__(self, n_channels: int, devices):
        self.L = _lib.load()
        self.n = int(n_channels)
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        check(self.L.hrfd_fanout_create(self.n, devs, len(devices), C.byref(h)), "hrfd_fanout_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.hrfd_fanout_destroy(self.h)
            self.h = None

    __del__ = close

    def set_mode(self, mode, channel=ALL):
        check(self.L.hrfd_fanout_set_mode(self.h, channel, mode), "hrfd_fanout_set_mode")

    def set_gain(self, mode, gain, channel=ALL):
        check(self.L.hrfd_fanout_set_gain(self.h, channel, mode, C.c_float(gain)), "hrfd_fanout_set_gain")

    def set_threshold(self, threshold, channel=ALL):
        check(self.L.hrfd_fanout_set_threshold(self.h, channel, threshold), "hrfd_fanout_set_threshold")

    def scatter(self, src_device: int, d_iq_all, block_bytes: int, n_blocks: int, src_stream=None):
        check(self.L.hrfd_fanout_scatter(self.h, int(src_device), _ptr(d_iq_all), int(block_bytes), int(n_blocks),
                                         _ptr(src_stream)), "hrfd_fanout_scatter")

    def process(self, gain_db: int = 0):
        check(self.L.hrfd_fanout_process(self.h, int(gain_db)), "hrfd_fanout_process")

    def collect(self, dst_device: int, d_pcm_all, d_n_pcm_all=None) -> int:
        """waits, repairs, gathers; returns the number of channels that were replayed on the exact path"""
        n = C.c_uint32(0)
        check(self.L.hrfd_fanout_collect(self.h, int(dst_device), _ptr(d_pcm_all), _ptr(d_n_pcm_all), C.byref(n)),
              "hrfd_fanout_collect")
        return int(n.value)


class Mod:
    """n_channels SSB modulators / interpolateSignal cascades (hrfd_mod_*)."""

    def __init__(self, kind: int, n_channels: int = 1, device: int = -1):
        self.L = _lib.load()
        self.n = int(n_channels)
        self.kind = kind
        h = C.c_void_p()
        check(self.L.hrfd_mod_create(kind, self.n, device, C.byref(h)), "hrfd_mod_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.hrfd_mod_destroy(self.h)
            self.h = None

    __del__ = close

    def reset(self, channel=ALL):
        check(self.L.hrfd_mod_reset(self.h, channel), "hrfd_mod_reset")

    def set_sideband(self, lsb, channel=ALL):
        check(self.L.hrfd_mod_set_sideband(self.h, channel, int(bool(lsb))), "hrfd_mod_set_sideband")

    def set_modulation_index(self, index, channel=ALL):
        check(self.L.hrfd_mod_set_modulation_index(self.h, channel, float(index)), "hrfd_mod_set_modulation_index")

    def set_deviation(self, deviation_hz, channel=ALL):
        check(self.L.hrfd_mod_set_deviation(self.h, channel, float(deviation_hz)), "hrfd_mod_set_deviation")

    def set_param(self, value, channel=ALL):
        """the kind's parameter: AM modulation index / FM, WBFM deviation (mirrors tests.reflib._Mod)"""
        (self.set_modulation_index if self.kind == MOD_AM else self.set_deviation)(value, channel)

    def process(self, pcm):
        """SSB / AM / FM: int16 [C, n]; INTERP: int16 [C, 2n] IQ pairs -> int8 [C, 512 n]"""
        pcm = np.ascontiguousarray(pcm, dtype=np.int16).reshape(self.n, -1)
        n = pcm.shape[1] // (2 if self.kind == MOD_INTERP else 1)
        out = np.zeros((self.n, 512 * n), dtype=np.int8)
        ob = C.c_uint32(0)
        check(self.L.hrfd_mod_process(self.h, _ptr(pcm), n, _ptr(out), C.byref(ob)), "hrfd_mod_process")
        assert ob.value == 512 * n
        return out if self.n > 1 else out[0]

    def debug_set_sliced(self, mode: int):
        """test hook (WBFM): 0 = the call's passes one after the other, 1 = time slices when the phase recurrence's
        stream has CUs of its own (the default), 2 = always"""
        check(self.L.hrfd_mod_debug_set_sliced(self.h, int(mode)), "hrfd_mod_debug_set_sliced")

    def debug_set_scan(self, kind: int):
        """test hook (FM, WBFM): 1 = the phase recurrence on round 2's k_phase_scan<64> (0: k_phase_rows, the default)"""
        check(self.L.hrfd_mod_debug_set_scan(self.h, int(kind)), "hrfd_mod_debug_set_scan")

    def debug_set_tail(self, kind: int):
        """test hook (WBFM): 0 = the lookup pass and the x8 cascade as two kernels (rounds 2-5), 1 = k_wb_tail (the default)"""
        check(self.L.hrfd_mod_debug_set_tail(self.h, int(kind)), "hrfd_mod_debug_set_tail")

    def process_device(self, d_pcm, n, d_out, stream=None):
        check(self.L.hrfd_mod_process_device(self.h, _ptr(d_pcm), n, _ptr(d_out), _ptr(stream)),
              "hrfd_mod_process_device")

    def sync(self):
        check(self.L.hrfd_mod_sync(self.h), "hrfd_mod_sync")


class Play:
    """hrfd_play_*: DataProvider's cyclic .iq playback, one read position per channel, image in HBM."""

    def __init__(self, n_channels: int = 1, device: int = -1):
        self.L = _lib.load()
        self.h = C.c_void_p()
        self.C = n_channels
        check(self.L.hrfd_play_create(n_channels, device, C.byref(self.h)), "hrfd_play_create")

    def close(self):
        if self.h:
            self.L.hrfd_play_destroy(self.h)
            self.h = C.c_void_p()

    __del__ = close

    def load_file(self, path: str):
        check(self.L.hrfd_play_load_file(self.h, path.encode()), "hrfd_play_load_file")

    def load(self, data: np.ndarray):
        data = np.ascontiguousarray(data, dtype=np.int8)
        check(self.L.hrfd_play_load(self.h, _ptr(data), data.size), "hrfd_play_load")

    def set_position(self, index: int, channel=ALL):
        check(self.L.hrfd_play_set_position(self.h, channel, index), "hrfd_play_set_position")

    def position(self, channel: int) -> int:
        v = C.c_uint32()
        check(self.L.hrfd_play_get_position(self.h, channel, C.byref(v)), "hrfd_play_get_position")
        return v.value

    def get(self, nbytes: int) -> np.ndarray:
        out = np.zeros((self.C, nbytes), dtype=np.int8)
        check(self.L.hrfd_play_get(self.h, _ptr(out), nbytes), "hrfd_play_get")
        return out

    def get_device(self, d_out, channel_stride: int, nbytes: int, stream=None):
        check(self.L.hrfd_play_get_device(self.h, d_out, channel_stride, nbytes, stream), "hrfd_play_get_device")


class Nco:
    """n_channels oscillators (hrfd_nco_*; Nco::run / Nco::runFast)."""

    def __init__(self, sample_rate: float, frequency: float, n_channels: int = 1, device: int = -1):
        self.L = _lib.load()
        self.n = int(n_channels)
        h = C.c_void_p()
        check(self.L.hrfd_nco_create(self.n, C.c_float(sample_rate), C.c_float(frequency), device, C.byref(h)),
              "hrfd_nco_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.hrfd_nco_destroy(self.h)
            self.h = None

    __del__ = close

    def set_frequency(self, f, channel=ALL):
        check(self.L.hrfd_nco_set_frequency(self.h, channel, C.c_float(f)), "hrfd_nco_set_frequency")

    def reset(self, channel=ALL):
        check(self.L.hrfd_nco_reset(self.h, channel), "hrfd_nco_reset")

    def run(self, count: int, fast: bool = False):
        i = np.zeros((self.n, count), dtype=np.float32)
        q = np.zeros((self.n, count), dtype=np.float32)
        check(self.L.hrfd_nco_run(self.h, int(fast), count, _ptr(i), _ptr(q)), "hrfd_nco_run")
        return (i, q) if self.n > 1 else (i[0], q[0])


class Engine:
    """Factory with the interface tests/goldencheck.py expects."""

    def ssbmod(self, lsb=True):
        m = Mod(MOD_SSB, 1)
        m.set_sideband(lsb)
        return m

    def interp(self):
        return Mod(MOD_INTERP, 1)

    def ammod(self):
        return Mod(MOD_AM, 1)

    def fmmod(self):
        return Mod(MOD_FM, 1)

    def wbfmmod(self):
        return Mod(MOD_WBFM, 1)

    def rx(self):
        return SingleChannelRx()

    def demod(self, mode):
        return Demod(mode, 1)


def q15_table(name: str) -> np.ndarray:
    L = _lib.load()
    buf = np.zeros(64, dtype=np.int16)
    n = L.hrfd_q15_table(name.encode(), buf.ctypes.data_as(C.POINTER(C.c_int16)), 64)
    return buf[:n].copy()


def atan2_table() -> np.ndarray:
    L = _lib.load()
    out = np.zeros((256, 256), dtype=np.float32)
    check(L.hrfd_atan2_table(out.ctypes.data_as(C.POINTER(C.c_float))))
    return out


def dbfs_table() -> np.ndarray:
    L = _lib.load()
    out = np.zeros(257, dtype=np.int32)
    check(L.hrfd_dbfs_table(out.ctypes.data_as(C.POINTER(C.c_int32))))
    return out


def device_count() -> int:
    return int(_lib.load().hrfd_device_count())
