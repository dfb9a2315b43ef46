// rx_demod_1024.hip -- instantiates rx_demod_kernel<1024, ...> (one translation unit per FFT size keeps the build parallel)
#include "rx_demod.hpp"
namespace ofdm {
hipError_t launch_rx_demod_1024(const RxDev& rx, const DemodArgs& a, hipStream_t s) { return launch_rx_demod_n<1024>(rx, a, s); }
}  // namespace ofdm
