// ecgpu_inst_msm.hip — instantiates the Pippenger pipeline for -DECGPU_CURVE=...
#include "ecgpu_msm.h"

namespace ecgpu {

using CurveT = ECGPU_CURVE;

template MsmPlan msm_plan<CurveT>(size_t n, int force_c);
template void launch_msm<CurveT>(const MsmPlan& p, hipStream_t s, const uint8_t* scalars, const uint8_t* xy, const uint8_t* inf,
                                 size_t n, void* workspace, uint32_t* out, int* status, hipEvent_t ev_sorted,
                                 hipEvent_t ev_accumulated);

}  // namespace ecgpu
