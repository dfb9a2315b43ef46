// rx_demod_512.hip -- instantiates rx_demod_kernel<512, ...> (one translation unit per FFT size keeps the build parallel)
#include "rx_demod.hpp"
namespace ofdm {
hipError_t launch_rx_demod_512(const RxDev& rx, const DemodArgs& a, hipStream_t s) { return launch_rx_demod_n<512>(rx, a, s); }
}  // namespace ofdm
