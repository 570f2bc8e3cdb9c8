// hackrfdiags_amd/csrc/hrfd_lib.hip -- unity translation unit of libhrfd.so.
// Kernels and their launchers live in one TU so that no relocatable device code
// / device link step is needed (plain `hipcc -c` + host link).
#include "hrfd_rx_kernels.hip"
#include "hrfd_rx_flow.hip"
#include "hrfd_rx_fir_kernels.hip"
#include "hrfd_rx_ragged.hip"
#include "../../include/hrfd.h"
#include "hrfd_tx_kernels.hip"
#include "hrfd_api.hip"
#include "hrfd_api_tx.hip"
#include "hrfd_api_debug.hip"
#include "hrfd_ingest.hip"
#include "hrfd_fanout.hip"
#include "hrfd_txring.hip"
#include "hrfd_play.hip"
#include "hrfd_membw.hip"
