// ecgpu_inst_msm.hip — instantiates the Pippenger pipeline for -DECGPU_CURVE=...
#include "ecgpu_msm.h"

namespace ecgpu {

using CurveT = ECGPU_CURVE;

template MsmPlan msm_plan<CurveT>(size_t n, int force_c, bool glv);
template bool msm_use_glv<CurveT>(size_t n);
template int msm_choose_window<CurveT>(size_t n);
template <> size_t msm_max_terms<CurveT>() { return (((size_t)1 << 31) - 64) / (MsmHasGlv<CurveT>::value ? 2 : 1); }
template void launch_msm<CurveT>(const MsmPlan& p, hipStream_t s, const uint8_t* scalars, const uint8_t* xy, const uint8_t* inf,
                                 size_t n, void* workspace, uint32_t* out, int* status, hipEvent_t ev_sorted,
                                 hipEvent_t ev_accumulated, uint8_t* out_xy, uint8_t* out_inf);
template void launch_msm_parts<CurveT>(const MsmPlan& p, hipStream_t s, const uint8_t* scalars, const uint8_t* xy, const uint8_t* inf,
                                       size_t n, void* workspace, uint32_t* parts, int* status, hipEvent_t ev_sorted,
                                       hipEvent_t ev_accumulated);
template void launch_msm_finish<CurveT>(const MsmPlan& p, hipStream_t s, const uint32_t* parts_all, int nranks, uint32_t* wins,
                                        uint32_t* out, uint8_t* out_xy, uint8_t* out_inf);

}  // namespace ecgpu
