// fused_sweep_f32.hip — fp32 instantiation of the fused sweep (armon_hip_sweep_f32; ref data_type=Float32).
#define ARMON_SWEEP_REAL float
#define ARMON_SWEEP_FN armon_hip_sweep_f32
#define ARMON_SWEEP_DESC armon_sweep_desc_f32
#define ARMON_TUNE_FN armon_hip_tune_placement_f32
#define ARMON_CHOOSE_FN armon_hip_choose_placement_f32
#include "fused_sweep_impl.hpp"
