/* oracle/hrfd_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C11) of the reference's 2.048 MS/s int8 IQ -> 8 kS/s
 * int16 PCM demodulation chain and its transmit mirror, written from the
 * closed-form stage equations of SURVEY.md section 8(a) -- block-vectorised,
 * no per-sample ring buffers.  Pinned bit-for-bit against the compiled
 * reference (oracle/_ref, see tests/test_oracle_vs_ref.py) and against the
 * committed golden vectors (tests/golden/).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * this library.  The product (hackrfdiags_amd/) never links or loads it.
 */
#ifndef HRFD_ORACLE_H
#define HRFD_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* IqDataProcessor::demodulatorType, hdr_diags/IqDataProcessor.h:21 */
enum { ORC_NONE = 0, ORC_AM = 1, ORC_FM = 2, ORC_WBFM = 3, ORC_LSB = 4, ORC_USB = 5 };

typedef struct orc_rx orc_rx;         /* IqDataProcessor + 4 demodulators + squelch */
typedef struct orc_demod orc_demod;   /* one {Am,Fm,WbFm,Ssb}Demodulator            */
typedef struct orc_ssbmod orc_ssbmod; /* SsbModulator                               */
typedef struct orc_ammod orc_ammod;   /* AmModulator                                */
typedef struct orc_fmmod orc_fmmod;   /* FmModulator                                */
typedef struct orc_wbfmmod orc_wbfmmod; /* WbFmModulator                            */
typedef struct orc_txring orc_txring; /* BasebandDataProcessor's PCM ring             */
typedef struct orc_interp orc_interp; /* signals/interpolateSignal cascade          */
typedef struct orc_nco orc_nco;       /* Nco + PhaseAccumulator                     */

/* ---- receive, outer boundary: IqDataProcessor::acceptIqData (IqDataProcessor.cc:926) */
orc_rx *orc_rx_create(void);
void orc_rx_destroy(orc_rx *h);
void orc_rx_set_mode(orc_rx *h, int mode);
void orc_rx_set_gain(orc_rx *h, int mode, float gain);
void orc_rx_set_threshold(orc_rx *h, int32_t threshold);
/* returns PCM sample count (0 when squelched / mode None). */
uint32_t orc_rx_process(orc_rx *h, const int8_t *iq, uint32_t bytes, uint32_t gain_db,
                        int16_t *pcm, uint32_t pcm_cap, uint32_t *magnitude,
                        int *signal_allowed, int8_t *iq256_out);
/* WBFM float stream of the last call (WbFmDemodulator::demodulatedData). */
uint32_t orc_rx_wbfm_float_stream(orc_rx *h, float *out, uint32_t cap);

/* ---- receive, inner boundary: X::acceptIqData(int8_t*,uint32_t) on 256 kS/s IQ */
orc_demod *orc_demod_create(int mode);
void orc_demod_destroy(orc_demod *h);
void orc_demod_reset(orc_demod *h);
void orc_demod_set_gain(orc_demod *h, float gain);
void orc_demod_set_sideband(orc_demod *h, int lsb);
uint32_t orc_demod_process(orc_demod *h, const int8_t *iq256, uint32_t bytes,
                           int16_t *pcm, uint32_t pcm_cap);

/* ---- transmit: SsbModulator::acceptData (SsbModulator.cc:455) */
orc_ssbmod *orc_ssbmod_create(int lsb);
void orc_ssbmod_destroy(orc_ssbmod *h);
void orc_ssbmod_reset(orc_ssbmod *h);
void orc_ssbmod_set_sideband(orc_ssbmod *h, int lsb);
uint32_t orc_ssbmod_process(orc_ssbmod *h, const int16_t *pcm, uint32_t n, int8_t *iq_out);

/* ---- transmit: signals/interpolateSignal.cc:250-374 (int16 IQ pairs -> int8 IQ x256) */
orc_ammod *orc_ammod_create(void);
void orc_ammod_destroy(orc_ammod *h);
void orc_ammod_reset(orc_ammod *h);
void orc_ammod_set_index(orc_ammod *h, float index);
uint32_t orc_ammod_process(orc_ammod *h, const int16_t *pcm, uint32_t n, int8_t *iq_out);

orc_fmmod *orc_fmmod_create(void);
void orc_fmmod_destroy(orc_fmmod *h);
void orc_fmmod_reset(orc_fmmod *h);
void orc_fmmod_set_deviation(orc_fmmod *h, float deviation);
uint32_t orc_fmmod_process(orc_fmmod *h, const int16_t *pcm, uint32_t n, int8_t *iq_out);

orc_wbfmmod *orc_wbfmmod_create(void);
void orc_wbfmmod_destroy(orc_wbfmmod *h);
void orc_wbfmmod_reset(orc_wbfmmod *h);
void orc_wbfmmod_set_deviation(orc_wbfmmod *h, float deviation);
uint32_t orc_wbfmmod_process(orc_wbfmmod *h, const int16_t *pcm, uint32_t n, int8_t *iq_out);

orc_txring *orc_txring_create(void);
void orc_txring_destroy(orc_txring *h);
void orc_txring_set_running(orc_txring *h, int running);
void orc_txring_write(orc_txring *h, const int16_t *pcm512);
void orc_txring_read(orc_txring *h, int16_t *pcm512);
void orc_txring_stats(const orc_txring *h, uint32_t *out6);

/* signals/{am,dsb,pm,fm}.cc: int16 PCM -> int16 (I,Q) pairs, the input of interpolateSignal.
 * kind 0 am (am.cc:40-52), 1 dsb (dsb.cc:38-46), 2 pm (pm.cc:41-53), 3 fm (fm.cc:44-77; *theta is
 * the tool's phase variable, 0 at program start and carried across calls). */
void orc_siggen(int kind, const int16_t *pcm, uint32_t n, int16_t *iq_pairs, float *theta);

orc_interp *orc_interp_create(void);
void orc_interp_destroy(orc_interp *h);
/* n_pairs IQ pairs in (2*n_pairs int16), 512*n_pairs bytes out */
uint32_t orc_interp_process(orc_interp *h, const int16_t *iq, uint32_t n_pairs, int8_t *iq_out);

/* ---- Nco (Nco/Nco.cc, Nco/PhaseAccumulator.cc) */
orc_nco *orc_nco_create(float sample_rate, float frequency);
void orc_nco_destroy(orc_nco *h);
void orc_nco_set_frequency(orc_nco *h, float frequency);
void orc_nco_reset(orc_nco *h);
void orc_nco_run(orc_nco *h, int fast, uint32_t count, float *i_out, float *q_out);
void orc_nco_tables(orc_nco *h, float *sin_out, float *cos_out);

/* ---- primitives and tables */
void orc_quantise(const float *coeffs, int count, int16_t *out);
uint32_t orc_decimate(const float *coeffs, int taps, int factor,
                      const int16_t *in, uint32_t count, int16_t *out);
void orc_interpolate(const float *coeffs, int taps, int factor,
                     const int16_t *in, uint32_t count, int16_t *out);
void orc_fir(const float *h, int n, const float *in, uint32_t count, float *out);
void orc_iir(const float *b, int nb, const float *a, int na,
             const float *in, uint32_t count, float *out);
int16_t orc_float_to_int16(float v);
void orc_atan2_lut(float *out /* [256][256], [q+128][i+128] */);
void orc_dbfs_table(int32_t *out /* [257] */);
/* named design tables for cross-checks: returns tap count, 0 if unknown */
int orc_table(const char *name, float *out, int cap);

#ifdef __cplusplus
}
#endif

#endif /* HRFD_ORACLE_H */
