// ecgpu_knobs.h — tuning knobs exist in the TOOL build of the library only.
//
// The product (lib/libecgpu.so) never reads the process environment: a drop-in for a `no_std`, `forbid(unsafe_code)` crate
// (k256/src/lib.rs:2,10) does not take its MSM plan from whoever set a variable in the process.  Every place that has a
// measured default and a sweepable alternative asks `knob("ECGPU_...")`; in the product that is a constant nullptr.
// `make -C elliptic-curves_amd` also links lib/libecgpu_knobs.so — the same objects, with the one translation unit that
// defines knob() (ecgpu_misc.hip) compiled -DECGPU_TUNING_KNOBS=1, where it is getenv — for tools/gpu_msm_*.py, the `env:`
// recipe of tools/gpu_run.sh, tools/gpu_fuzz.py and the few tests that force a code path (two-level sort at small n, chunk
// sizes, the chunked host-pointer MSM, the fused tail): they load it through ECGPU_TOOL_LIB / Engine(..., variant="knobs").
// Ranges are validated where a knob is read; results never depend on one.
#pragma once

namespace ecgpu {
const char* knob(const char* name);
}
