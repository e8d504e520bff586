// Launchers not yet backed by kernels in this build (calls are rejected in rg_batch_create).
#include "rg_poa_args.hpp"
namespace rg {
void launch_m2(const PoaArgs&, hipStream_t) {}
void launch_m0_scalar(const PoaArgs&, hipStream_t) {}
}  // namespace rg
