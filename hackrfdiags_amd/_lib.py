"""Locate / load libhrfd.so (the C ABI of include/hrfd.h) with ctypes.

There is deliberately no fallback: if the shared library is missing the import
raises, and if there is no GPU every create call returns HRFD_ENODEV.
"""
from __future__ import annotations

import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# HRFD_LIB selects another build of the same library (tools/gpu_ab.py compares tuning variants)
LIB_PATH = os.environ.get("HRFD_LIB") or os.path.join(PKG_DIR, "lib", "libhrfd.so")

_i16p = C.POINTER(C.c_int16)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)
_vp = C.c_void_p

_lib = None
_runtime = None        # the HIP runtime CDLL the library was bound to


def _load_hip_runtime():
    """libhrfd.so carries no DT_NEEDED entry for the HIP runtime: a process must
    hold exactly ONE HIP/HSA runtime.  When PyTorch is importable we bind to the
    runtime it ships (so torch tensors, streams and RCCL share the device context
    with our kernels); otherwise to the system ROCm runtime.  Set
    HRFD_HIP_RUNTIME=/path/libamdhip64.so to force a choice."""
    global _runtime
    if _runtime is not None:
        return _runtime
    cands = []
    forced = os.environ.get("HRFD_HIP_RUNTIME")
    if forced:
        cands.append(forced)
    elif os.environ.get("HRFD_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401  (loads its bundled runtime)
            tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
            if os.path.exists(tl):
                cands.append(tl)
        except Exception:
            pass
    cands += ["/opt/rocm/lib/libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so"]
    last = None
    for c in cands:
        try:
            _runtime = C.CDLL(c, mode=C.RTLD_GLOBAL)
            return _runtime
        except OSError as e:      # try the next candidate
            last = e
    raise HrfdError(f"no HIP runtime could be loaded ({last}); libhrfd.so needs libamdhip64")


class HrfdError(RuntimeError):
    pass


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HrfdError(f"{LIB_PATH} not found: build it with "
                        "`python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    _load_hip_runtime()
    L = C.CDLL(LIB_PATH)
    L.hrfd_last_error.restype = C.c_char_p
    L.hrfd_version.restype = C.c_int
    L.hrfd_device_count.restype = C.c_int
    L.hrfd_rx_create.argtypes = [C.c_uint32, C.c_int, C.POINTER(_vp)]
    L.hrfd_rx_destroy.argtypes = [_vp]
    L.hrfd_rx_set_mode.argtypes = [_vp, C.c_uint32, C.c_int]
    L.hrfd_rx_set_gain.argtypes = [_vp, C.c_uint32, C.c_int, C.c_float]
    L.hrfd_rx_set_threshold.argtypes = [_vp, C.c_uint32, C.c_int32]
    L.hrfd_rx_reset_demod.argtypes = [_vp, C.c_uint32, C.c_int]
    L.hrfd_rx_process_block.argtypes = [_vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32,
                                        _vp, _vp, _vp, _vp, _vp]
    L.hrfd_rx_process_device.argtypes = [_vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                         _vp, _vp, _vp, _vp, _vp, _vp]
    L.hrfd_rx_sync.argtypes = [_vp, _u32p]
    for name in ("hrfd_rx_pcm_capacity", "hrfd_rx_iq256_capacity", "hrfd_demod_pcm_capacity"):
        getattr(L, name).argtypes = [C.c_uint32]
        getattr(L, name).restype = C.c_uint32
    L.hrfd_rx_pending_samples.argtypes = [_vp, _u32p]
    L.hrfd_rx_debug_ragged.argtypes = [_vp, _i32p, C.POINTER(C.c_ulonglong)]
    L.hrfd_rx_reduce_sample_rate.argtypes = [_vp, _vp, C.c_uint32, _vp]
    L.hrfd_rx_failed_channels.argtypes = [_vp, C.c_void_p, C.c_uint32]
    L.hrfd_rx_debug_set_warm.argtypes = [_vp, C.c_int]
    L.hrfd_rx_debug_set_atan.argtypes = [_vp, C.c_int]
    L.hrfd_rx_debug_atan_eval.argtypes = [_vp, _f32p]
    L.hrfd_rx_debug_atan_eval_tab.argtypes = [_vp, _f32p]
    for name, args in (("hrfd_rx_debug_atan_eval_quad", [_vp, _f32p]), ("hrfd_debug_atan2_quadrant", [_u32p, _i32p])):
        if hasattr(L, name):                               # (round-5 hooks: an older build named by HRFD_LIB, as tools/flow_ab.sh compares, has none)
            getattr(L, name).argtypes = args
    L.hrfd_rx_debug_counters.argtypes = [_vp, _u32p]
    L.hrfd_rx_debug_set_stagger.argtypes = [_vp, C.c_int]
    L.hrfd_rx_debug_expire.argtypes = [_vp, C.c_int]
    L.hrfd_rx_debug_set_gated.argtypes = [_vp, C.c_int]
    L.hrfd_rx_debug_set_fir_flow.argtypes = [_vp, C.c_int]
    L.hrfd_rx_debug_set_stream.argtypes = [_vp, C.c_int]
    L.hrfd_rx_debug_set_run_len.argtypes = [_vp, C.c_int]
    L.hrfd_rx_debug_stamps.argtypes = [_vp, C.c_uint32, _vp]
    L.hrfd_rx_debug_enable_timing.argtypes = [_vp, C.c_int]
    L.hrfd_rx_debug_kernel_ms.argtypes = [_vp, C.c_int, _f32p]
    L.hrfd_rx_debug_timing_every.argtypes = [_vp, C.c_int]
    L.hrfd_debug_membw.argtypes = [C.c_int, _vp, C.c_size_t, _vp, _vp]
    L.hrfd_demod_create.argtypes = [C.c_int, C.c_uint32, C.c_int, C.POINTER(_vp)]
    L.hrfd_demod_destroy.argtypes = [_vp]
    L.hrfd_demod_reset.argtypes = [_vp, C.c_uint32]
    L.hrfd_demod_set_gain.argtypes = [_vp, C.c_uint32, C.c_float]
    L.hrfd_demod_set_sideband.argtypes = [_vp, C.c_uint32, C.c_int]
    L.hrfd_demod_process.argtypes = [_vp, _vp, C.c_uint32, _vp, _vp]
    L.hrfd_ingest_create.argtypes = [_vp, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_vp)]
    L.hrfd_ingest_destroy.argtypes = [_vp]
    L.hrfd_ingest_acquire.argtypes = [_vp, C.POINTER(_vp)]
    L.hrfd_ingest_submit.argtypes = [_vp, C.c_uint32]
    L.hrfd_ingest_collect.argtypes = [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]
    L.hrfd_ingest_replayed.argtypes = [_vp, C.POINTER(C.c_uint64)]
    L.hrfd_fanout_channel_range.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, _u32p, _u32p]
    L.hrfd_fanout_create.argtypes = [C.c_uint32, C.POINTER(C.c_int), C.c_uint32, C.POINTER(_vp)]
    L.hrfd_fanout_destroy.argtypes = [_vp]
    L.hrfd_fanout_shards.argtypes = [_vp, _u32p]
    L.hrfd_fanout_set_mode.argtypes = [_vp, C.c_uint32, C.c_int]
    L.hrfd_fanout_set_gain.argtypes = [_vp, C.c_uint32, C.c_int, C.c_float]
    L.hrfd_fanout_set_threshold.argtypes = [_vp, C.c_uint32, C.c_int32]
    L.hrfd_fanout_scatter.argtypes = [_vp, C.c_int, _vp, C.c_uint32, C.c_uint32, _vp]
    L.hrfd_fanout_input.argtypes = [_vp, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_vp), _u32p, _u32p]
    L.hrfd_fanout_process.argtypes = [_vp, C.c_uint32]
    L.hrfd_fanout_collect.argtypes = [_vp, C.c_int, _vp, _vp, _u32p]
    L.hrfd_txring_create.argtypes = [C.c_uint32, C.POINTER(_vp)]
    L.hrfd_txring_destroy.argtypes = [_vp]
    L.hrfd_txring_set_running.argtypes = [_vp, C.c_uint32, C.c_int]
    L.hrfd_txring_write.argtypes = [_vp, C.c_uint32, _vp]
    L.hrfd_txring_read_batch.argtypes = [_vp, _vp]
    L.hrfd_txring_stats.argtypes = [_vp, C.c_uint32, _u32p]
    L.hrfd_mod_create.argtypes = [C.c_int, C.c_uint32, C.c_int, C.POINTER(_vp)]
    L.hrfd_mod_destroy.argtypes = [_vp]
    L.hrfd_mod_reset.argtypes = [_vp, C.c_uint32]
    L.hrfd_mod_set_sideband.argtypes = [_vp, C.c_uint32, C.c_int]
    L.hrfd_mod_set_modulation_index.argtypes = [_vp, C.c_uint32, C.c_float]
    L.hrfd_mod_set_deviation.argtypes = [_vp, C.c_uint32, C.c_float]
    L.hrfd_mod_process.argtypes = [_vp, _vp, C.c_uint32, _vp, _u32p]
    L.hrfd_mod_process_device.argtypes = [_vp, _vp, C.c_uint32, _vp, _vp]
    L.hrfd_mod_sync.argtypes = [_vp]
    if hasattr(L, "hrfd_libm_variant"):
        L.hrfd_libm_variant.argtypes = []
        L.hrfd_libm_variant.restype = C.c_int
    L.hrfd_mod_debug_set_sliced.argtypes = [_vp, C.c_int]
    L.hrfd_mod_debug_set_scan.argtypes = [_vp, C.c_int]
    if hasattr(L, "hrfd_mod_debug_set_tail"):              # (round 6; an older build named by HRFD_LIB has none)
        L.hrfd_mod_debug_set_tail.argtypes = [_vp, C.c_int]
    L.hrfd_play_create.argtypes = [C.c_uint32, C.c_int, C.POINTER(_vp)]
    L.hrfd_play_destroy.argtypes = [_vp]
    L.hrfd_play_load_file.argtypes = [_vp, C.c_char_p]
    L.hrfd_play_load.argtypes = [_vp, _vp, C.c_uint32]
    L.hrfd_play_set_position.argtypes = [_vp, C.c_uint32, C.c_uint32]
    L.hrfd_play_get_position.argtypes = [_vp, C.c_uint32, _u32p]
    L.hrfd_play_get_device.argtypes = [_vp, _vp, C.c_uint64, C.c_uint32, _vp]
    L.hrfd_play_get.argtypes = [_vp, _vp, C.c_uint32]
    L.hrfd_nco_create.argtypes = [C.c_uint32, C.c_float, C.c_float, C.c_int, C.POINTER(_vp)]
    L.hrfd_nco_destroy.argtypes = [_vp]
    L.hrfd_nco_set_frequency.argtypes = [_vp, C.c_uint32, C.c_float]
    L.hrfd_nco_reset.argtypes = [_vp, C.c_uint32]
    L.hrfd_nco_run.argtypes = [_vp, C.c_int, C.c_uint32, _vp, _vp]
    L.hrfd_q15_table.argtypes = [C.c_char_p, _i16p, C.c_int]
    L.hrfd_atan2_table.argtypes = [_f32p]
    L.hrfd_dbfs_table.argtypes = [_i32p]
    _lib = L
    return L


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().hrfd_last_error().decode(errors="replace")
        raise HrfdError(f"{what} failed ({rc}): {msg}")
