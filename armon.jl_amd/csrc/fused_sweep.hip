// fused_sweep.hip — placeholder until the fused kernels land (next commit).
#include "common.hpp"
using namespace armon;
extern "C" int armon_hip_sweep(armon_ctx* ctx, const armon_sweep_desc* d)
{
    ARMON_REQUIRE(ctx && d, "NULL argument");
    ARMON_REQUIRE(false, "fused sweep not built yet");
}
