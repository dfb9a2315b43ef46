// rx_demod_2048.hip -- instantiates rx_demod_kernel<2048, ...> (one translation unit per FFT size keeps the build parallel)
#include "rx_demod.hpp"
namespace ofdm {
hipError_t launch_rx_demod_2048(const RxDev& rx, const DemodArgs& a, hipStream_t s) { return launch_rx_demod_n<2048>(rx, a, s); }
}  // namespace ofdm
