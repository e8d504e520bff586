// Pathwise driver (placeholder until the kernels land in this file).
#include "rg_path_args.hpp"
namespace rg {
struct PathWorkImpl {};
PathWork::~PathWork() { delete impl; }
int path_driver_run(const HostGraph&, const PathGraphDev&, const rg_params&, PathWork&, const uint8_t*, const long long*,
                    const uint8_t*, int, int, DevRecord*, uint8_t*, long long, unsigned long long*, hipStream_t,
                    std::vector<std::pair<std::string, std::pair<double, long long>>>&) {
    return fail(RG_ERR_ARG, "pathwise modes are not built into this library yet");
}
void launch_m2(const struct PoaArgs&, hipStream_t) {}
void launch_m0_scalar(const struct PoaArgs&, hipStream_t) {}
}  // namespace rg
