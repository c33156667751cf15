// ecgpu_inst_var.hip — instantiates the variable-base kernel for -DECGPU_CURVE=...
#include "ecgpu_launch.h"
#include "ecgpu_var.h"

namespace ecgpu {

using CurveT = ECGPU_CURVE;

// Table slots = resident lanes: at most 256 CUs x 8 workgroups of 256; lanes stride over the batch.
template <> size_t var_base_slots<CurveT>(size_t n) {
    size_t slots = ((n + BLOCK - 1) / BLOCK) * BLOCK;
    const size_t max_slots = (size_t)256 * 8 * BLOCK;
    return slots > max_slots ? max_slots : slots;
}
template <> size_t var_base_tab_words<CurveT>() { return (size_t)8 * VAR_TAB_ELEMS * CurveT::NL; }
template <> void launch_var_base<CurveT>(hipStream_t s, const uint8_t* scalars, const uint8_t* xy, const uint8_t* inf, size_t n,
                                         uint32_t* tab, size_t slots, uint32_t* proj_out, int* status, uint32_t* add_io) {
    if (add_io)
        hipLaunchKernelGGL((k_var_base<CurveT, true>), dim3((unsigned)(slots / BLOCK)), dim3(BLOCK), 0, s, scalars, xy, inf, n, tab, slots,
                           proj_out, status, add_io);
    else
        hipLaunchKernelGGL((k_var_base<CurveT, false>), dim3((unsigned)(slots / BLOCK)), dim3(BLOCK), 0, s, scalars, xy, inf, n, tab, slots,
                           proj_out, status, add_io);
}

}  // namespace ecgpu
