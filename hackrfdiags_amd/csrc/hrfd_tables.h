// hackrfdiags_amd/csrc/hrfd_tables.h
//
// Q15 filter taps used by the HIP kernels, stored ALREADY QUANTISED: these are
// the int16 values the reference constructors derive at start-up with
// (int16_t)round(h*32768) (Filters/Int16/Decimator_int16.cc:55-63).  The float
// design values live only in the test oracle; tests/test_tables.py checks that
// quantising them gives exactly these integers.
#ifndef HRFD_TABLES_H
#define HRFD_TABLES_H

#include <stdint.h>

namespace hrfd {

// IqDataProcessor.cc:8-13 (front end stage 1); SsbModulator interpolator 8
static constexpr int N_HB1 = 3;
static constexpr int16_t Q_HB1[3] = {
  8206, 16384, 8206
};

// IqDataProcessor.cc:15-20 (front end stage 2); SsbModulator interpolator 7
static constexpr int N_HB2 = 3;
static constexpr int16_t Q_HB2[3] = {
  8249, 16384, 8249
};

// IqDataProcessor.cc:22-27 (front end stage 3); SsbModulator interpolators 3,6
static constexpr int N_HB3 = 3;
static constexpr int16_t Q_HB3[3] = {
  8424, 16384, 8424
};

// WbFmDemodulator.cc:16-26
static constexpr int N_WBFM_D1 = 8;
static constexpr int16_t Q_WBFM_D1[8] = {
  799, 2522, 4796, 6446, 6446, 4796, 2522, 799
};

// WbFmDemodulator.cc:28-42, FmDemodulator.cc:53-67
static constexpr int N_POST_D12 = 12;
static constexpr int16_t Q_POST_D12[12] = {
  75, 777, 1984, 3693, 5391, 6459, 6459, 5391, 3693, 1984,
  777, 75
};

// WbFmDemodulator.cc:44-86, FmDemodulator.cc:69-111, SsbModulator.cc:13-55
static constexpr int N_AUDIO_D40 = 40;
static constexpr int16_t Q_AUDIO_D40[40] = {
  52, -364, -886, -870, -76, 592, 215, -601, -437, 605,
  757, -528, -1192, 299, 1803, 230, -2826, -1631, 5877, 13585,
  13585, 5877, -1631, -2826, 230, 1803, 299, -1192, -528, 757,
  605, -437, -601, 215, 592, -76, -870, -886, -364, 52
};

// FmDemodulator.cc:17-51
static constexpr int N_FM_TUNER_D32 = 32;
static constexpr int16_t Q_FM_TUNER_D32[32] = {
  135, 178, 249, 378, 497, 666, 824, 1020, 1201, 1400,
  1575, 1748, 1887, 2005, 2082, 2124, 2124, 2082, 2005, 1887,
  1748, 1575, 1400, 1201, 1020, 824, 666, 497, 378, 249,
  178, 135
};

// AmDemodulator.cc:14-24, SsbDemodulator.cc:14-24
static constexpr int N_AM_D1 = 8;
static constexpr int16_t Q_AM_D1[8] = {
  795, 2511, 4776, 6419, 6419, 4776, 2511, 795
};

// AmDemodulator.cc:26-40, SsbDemodulator.cc:26-40
static constexpr int N_AM_D2 = 12;
static constexpr int16_t Q_AM_D2[12] = {
  188, 865, 1983, 3521, 4992, 5914, 5914, 4992, 3521, 1983,
  865, 188
};

// AmDemodulator.cc:42-62, SsbDemodulator.cc:42-62
static constexpr int N_AM_D3 = 16;
static constexpr int16_t Q_AM_D3[16] = {
  382, 500, -360, -2005, -2412, 615, 6515, 11408, 11408, 6515,
  615, -2412, -2005, -360, 500, 382
};

// SsbDemodulator.cc delayLineCoefficients, SsbModulator.cc:126 (1.0 quantises to -32768)
static constexpr int N_SSB_DELAY = 16;
static constexpr int16_t Q_SSB_DELAY[16] = {
  0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
  0, 0, 0, 0, 0, -32768
};

// SsbDemodulator.cc phaseShifterCoefficients, SsbModulator.cc:128-162
static constexpr int N_SSB_HILBERT = 31;
static constexpr int16_t Q_SSB_HILBERT[31] = {
  -111, 0, -192, 0, -440, 0, -922, 0, -1753, 0,
  -3213, 0, -6343, 0, -20651, 0, 20651, 0, 6343, 0,
  3213, 0, 1753, 0, 922, 0, 440, 0, 192, 0,
  111
};

// SsbModulator.cc interpolators 2,4,5; interpolateSignal.cc stages 2,4,5
static constexpr int N_INTERP_HB8 = 8;
static constexpr int16_t Q_INTERP_HB8[8] = {
  -1445, 0, 9548, 16384, 9548, 0, -1445, 0
};

// SsbModulator.cc interpolators 3,6
static constexpr int N_INTERP_HB3 = 4;
static constexpr int16_t Q_INTERP_HB3[4] = {
  8424, 16384, 8424, 0
};

// SsbModulator.cc interpolator 7
static constexpr int N_INTERP_HB2 = 4;
static constexpr int16_t Q_INTERP_HB2[4] = {
  8249, 16384, 8249, 0
};

// SsbModulator.cc interpolator 8
static constexpr int N_INTERP_HB1 = 4;
static constexpr int16_t Q_INTERP_HB1[4] = {
  8206, 16384, 8206, 0
};

// signals/interpolateSignal.cc:30-72 (asymmetric stage 1)
static constexpr int N_INTERPSIG_S1 = 40;
static constexpr int16_t Q_INTERPSIG_S1[40] = {
  -37, 601, 100, -328, -194, 378, 358, -396, -576, 362,
  861, -245, -1237, -10, 1773, 544, -2717, 1925, 5691, 13835,
  13835, 5691, -1925, -2717, 544, 1773, -10, -1237, -245, 861,
  362, -576, -396, 358, 378, -194, -328, 100, 601, -37
};

// float constants of the recursive sections (exact float literals of the reference)
static constexpr float DEEMPH_B0 = 0.0253863f;   // WbFmDemodulator.cc:93-97 (b0 == b1)
static constexpr float DEEMPH_A1 = -0.9492274f;  // WbFmDemodulator.cc:99-102
static constexpr float DCREM_A1 = -0.95f;        // AmDemodulator.cc:68, SsbDemodulator.cc (b = {1,-1})

struct NamedTable { const char *name; const int16_t *taps; int n; };
static const NamedTable kNamedTables[] = {
  {"HB1", Q_HB1, N_HB1},
  {"HB2", Q_HB2, N_HB2},
  {"HB3", Q_HB3, N_HB3},
  {"WBFM_D1", Q_WBFM_D1, N_WBFM_D1},
  {"POST_D12", Q_POST_D12, N_POST_D12},
  {"AUDIO_D40", Q_AUDIO_D40, N_AUDIO_D40},
  {"FM_TUNER_D32", Q_FM_TUNER_D32, N_FM_TUNER_D32},
  {"AM_D1", Q_AM_D1, N_AM_D1},
  {"AM_D2", Q_AM_D2, N_AM_D2},
  {"AM_D3", Q_AM_D3, N_AM_D3},
  {"SSB_DELAY", Q_SSB_DELAY, N_SSB_DELAY},
  {"SSB_HILBERT", Q_SSB_HILBERT, N_SSB_HILBERT},
  {"INTERP_HB8", Q_INTERP_HB8, N_INTERP_HB8},
  {"INTERP_HB3", Q_INTERP_HB3, N_INTERP_HB3},
  {"INTERP_HB2", Q_INTERP_HB2, N_INTERP_HB2},
  {"INTERP_HB1", Q_INTERP_HB1, N_INTERP_HB1},
  {"INTERPSIG_S1", Q_INTERPSIG_S1, N_INTERPSIG_S1},
};

} // namespace hrfd

#endif // HRFD_TABLES_H
