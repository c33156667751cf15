// ecgpu_inst_msmtail.hip — the latency-bound tail of the Pippenger pipeline (bucket finish, running sums, trees over the
// segment sums, window sums, the Horner chain of doublings) for -DECGPU_CURVE=..., built with
// `-mllvm -amdgpu-sched-strategy=max-ilp` (Makefile): these kernels run ONE wave per SIMD, each lane walking a chain of
// dependent point operations, so what they wait for is their own instruction latency — the scheduler is asked to interleave
// the independent multiplications of a point operation instead of minimising registers for an occupancy they never reach.
// Variant tag 1 of the kernels (ecgpu_msm.h); selected with ECGPU_MSM_TAIL.
#include "ecgpu_msm.h"

namespace ecgpu {

using CurveT = ECGPU_CURVE;

template void launch_msm_tail<CurveT, 1>(const MsmPlan& p, hipStream_t stream, uint8_t* ws, uint32_t* parts, hipEvent_t ev_accumulated);
template void launch_msm_finish_v<CurveT, 1>(const MsmPlan& p, hipStream_t stream, const uint32_t* parts_all, int nranks, uint32_t* wins,
                                             uint32_t* out);

}  // namespace ecgpu
