// fused_sweep_f64.hip — fp64 instantiation of the fused sweep (armon_hip_sweep).
#define ARMON_SWEEP_REAL double
#define ARMON_SWEEP_FN armon_hip_sweep
#define ARMON_SWEEP_DESC armon_sweep_desc
#define ARMON_TUNE_FN armon_hip_tune_placement
#define ARMON_CHOOSE_FN armon_hip_choose_placement
#define ARMON_CYCLE_FN armon_hip_cycle_xy
#include "fused_sweep_impl.hpp"
