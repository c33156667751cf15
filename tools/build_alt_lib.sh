#!/bin/bash
# A second build of libecgpu.so for A/B measurements (the `ab:` recipe of tools/gpu_run.sh loads it through ECGPU_TOOL_LIB): the named
# translation units are recompiled with extra flags into elliptic-curves_amd/build_alt/, everything else is linked from the main
# build.      bash tools/build_alt_lib.sh <suffix> "<extra hipcc flags>" <group>_<Curve> [...]
#   bash tools/build_alt_lib.sh nofused "-DECGPU_FUSED_SUB=0" var_P256Params var_P384Params        -> lib/libecgpu_nofused.so
#   bash tools/build_alt_lib.sh acc4 "-DECGPU_MSM_ACC_WAVES=4" msm_K256Params                      -> lib/libecgpu_acc4.so
# (the alternative libraries are build artefacts: git-ignored, they travel to the GPU box with the snapshot)
set -eu
cd "$(dirname "$0")/../elliptic-curves_amd"
SUFFIX=${1:?suffix}; FLAGS=${2:?flags}; shift 2
make -j"$(nproc)" > /dev/null
mkdir -p build_alt
SKIP=""
for tu in "$@"; do
  group=${tu%%_*}; curve=${tu#*_}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $FLAGS -DECGPU_CURVE="$curve" -c "csrc/ecgpu_inst_$group.hip" -o "build_alt/inst_${group}_$curve.o" &
  SKIP="$SKIP|inst_${group}_$curve.o"
done
wait
OBJS=$(ls build/*.o | grep -Ev "${SKIP#|}|misc_knobs.o")      # (misc_knobs.o belongs to the tool build lib/libecgpu_knobs.so only)
ALT=$(for tu in "$@"; do echo "build_alt/inst_${tu%%_*}_${tu#*_}.o"; done)
hipcc --offload-arch=gfx950 -shared -fPIC -o "lib/libecgpu_$SUFFIX.so" $OBJS $ALT -ldl -lpthread
echo "lib/libecgpu_$SUFFIX.so"
