// pirip_amd/csrc/fsk_demod_fast.hip -- specialised kernel for the headline configuration.
// (placeholder until the general kernel is parity-green on hardware)
#include <hip/hip_runtime.h>
#include "fsk_device.hpp"
namespace pirip {
bool demod_fast_applicable(const FskDims &) { return false; }
hipError_t launch_demod_fast(const DemodArgs &, int, hipStream_t) { return hipErrorNotSupported; }
}  // namespace pirip
