/* include/hrfd_debug.h -- the hrfd_*_debug_* entry points of libhrfd.so: NOT part of the drop-in boundary
 * (include/hrfd.h is; nothing here stands in for an interface of the reference).  They exist for the repository's
 * own tests, tools and bench.py.  Two kinds:
 *
 *   read-only introspection / measurement -- always available
 *   behaviour-changing test hooks         -- INERT unless the process was started with HRFD_DEBUG_HOOKS=1 in its
 *                                            environment (read once, at the first such call): without it they return
 *                                            HRFD_ESTATE and change nothing.  tests/conftest.py sets it; a host
 *                                            application never does, so nothing can switch the shipped library onto its
 *                                            forced-failure and fallback paths at run time.
 */
#ifndef HRFD_DEBUG_H
#define HRFD_DEBUG_H

#include "hrfd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ read-only introspection / measurement */
/* {launches, ..., [4] tiles repaired in place, [5] launches with uncommitted channels, [6] launches} of the handle */
int hrfd_rx_debug_counters(hrfd_rx *h, uint32_t *out8);
/* bracket the demodulator kernels of every launch with HIP events on the launch stream (`slots` pairs, round robin);
 * after a sync hrfd_rx_debug_kernel_ms(slot) is that launch's time: bench.py's roofline.achieved */
int hrfd_rx_debug_enable_timing(hrfd_rx *h, int slots);
int hrfd_rx_debug_kernel_ms(hrfd_rx *h, int slot, float *ms);
/* bracket only every n-th launch (an event record costs ~3 us of queue time: bracketing every launch of a back-to-back
 * sequence puts a gap between kernels that a host which does not measure never sees) */
int hrfd_rx_debug_timing_every(hrfd_rx *h, int n);
/* the device's atan2 (arithmetic form / first-octant-table form) for all 65536 (q, i): must equal hrfd_atan2_table() */
int hrfd_rx_debug_atan_eval(hrfd_rx *h, float *out65536);
int hrfd_rx_debug_atan_eval_tab(hrfd_rx *h, float *out65536);
int hrfd_rx_debug_atan_eval_quad(hrfd_rx *h, float *out65536);   /* first-quadrant table (the re-split WBFM flow kernel) */
/* host only: that table as hrfd_rx_create builds it (16644 words) and the verdict of its proof against hrfd_atan2_table() */
int hrfd_debug_atan2_quadrant(uint32_t *out16644, int *ok);
/* per-workgroup cycle stamps of k_rx_wbfm (probe builds); the cross-block check values of the latest launch */
int hrfd_rx_debug_stamps(hrfd_rx *h, uint32_t cap_groups, unsigned long long *host_out);
int hrfd_rx_debug_chk(hrfd_rx *h, float *pub, float *spec, uint32_t n);
/* *offgrid: the handle was given a block that is not a whole number of PCM samples (512 bytes; inner API 64) and keeps
 * its state in RagState since; *launches: launches so far that ran on k_rx_ragged (hrfd_rx_ragged.hip) */
int hrfd_rx_debug_ragged(hrfd_rx *h, int *offgrid, unsigned long long *launches);

/* two plain stream kernels -- kind 0 reads `bytes` of d_buf (d_sink: one dword that is never written, may be NULL),
 * kind 1 overwrites them; bytes a multiple of 256 KiB; asynchronous on `stream`.  bench.py times them in the same run
 * as the measured denominators beside the 8 TB/s spec peak (`measured_stream_read_GBps` / `_write_GBps`). */
int hrfd_debug_membw(int kind, void *d_buf, size_t bytes, void *d_sink, void *stream);

/* ------------------------------------------------------------------ behaviour-changing hooks (HRFD_DEBUG_HOOKS=1) */
int hrfd_rx_debug_set_atan(hrfd_rx *h, int mode);        /* -1 automatic, 0 table gather, 1 arithmetic */
int hrfd_rx_debug_set_warm(hrfd_rx *h, int warm);        /* shrink the de-emphasis warm-up, seeds off: forces repairs */
int hrfd_rx_debug_set_run_len(hrfd_rx *h, int blocks);   /* blocks per k_rx_wbfm workgroup */
int hrfd_rx_debug_set_stream(hrfd_rx *h, int on);        /* 0: WBFM batches on the block kernel */
int hrfd_rx_debug_expire(hrfd_rx *h, int where);         /* expire a bounded wait / hold a wave up, once */
int hrfd_rx_debug_set_fir_flow(hrfd_rx *h, int mode);    /* FIR modes: -1 automatic, 0 block kernels, 1 flow, 2 per kind */
int hrfd_rx_debug_set_gated(hrfd_rx *h, int on);         /* 0: no gated pass on the device (host replay instead) */
int hrfd_rx_debug_set_stagger(hrfd_rx *h, int units);
int hrfd_mod_debug_set_sliced(hrfd_mod *h, int on);      /* WBFM modulator: 0 unsliced, 1 automatic, 2 always sliced */
int hrfd_mod_debug_set_scan(hrfd_mod *h, int kind);      /* FM / WBFM modulators: the phase recurrence on 0 = k_phase_rows8 / k_phase_rows, 1 = k_phase_scan<64>, 2 = k_phase_rows */
int hrfd_mod_debug_set_tail(hrfd_mod *h, int kind);      /* WBFM modulator: 0 = k_wb_rails + k_mod<WB_TAIL> (rounds 2-5), 1 = k_wb_tail (one pass) */

#ifdef __cplusplus
}
#endif

#endif /* HRFD_DEBUG_H */
