/* ofdm_experiments.h -- bench-only additions to the C ABI, exported ONLY by tools/experiments/libofdm_mi355x_exp.so
 * (the product sources compiled with -DOFDM_EXPERIMENTS: `make -C lte-gnu-radio-code_amd/csrc exp`).
 * The product library libofdm_mi355x.so neither exports these symbols nor contains the kernels they select.
 * Used by tools/kbench.py (interleaved A/B timing of kernel variants) and tools/stamps.py (per-phase cycle stamps). */
#ifndef OFDM_EXPERIMENTS_H
#define OFDM_EXPERIMENTS_H
#include "../../include/ofdm_mi355x.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Select an alternative build of rx_demod_kernel<2048, 16-QAM, packed bits> for the handle's batch path.  0 = the shipped
 * kernel.  Variants are documented where they are instantiated (csrc/rx_demod.hpp, launch_rx_demod_n); some exist only to
 * time a hypothesis and do NOT produce the product's results (kbench checks outputs unless told otherwise).
 * >= 100: shipped kernel with (variant-100) KiB of unused LDS added to the launch (occupancy experiment). */
int ofdm_exp_set_variant(ofdm_rx* h, int32_t variant);
/* Device buffer of 8 uint32 per wave receiving per-phase cycle sums (s_memtime stamps) of the stamped variant. */
int ofdm_exp_set_stamp_buffer(ofdm_rx* h, void* d_stamps);
#ifdef __cplusplus
}
#endif
#endif
