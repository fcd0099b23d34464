#!/bin/bash
# build_variant.sh NAME [extra hipcc flags]: conv3d.hip / conv_march.hip / volume.hip / wgrad.hip (+ tools/experiments/conv_pipe.hip, the persistent LDS-DMA
# form that only the -DOSA_EXPERIMENTS build links) compiled with extra flags (e.g. -DOSA_EXPERIMENTS: the measurement switches of
# osa_common.h), linked with the regular objects into openstereo_amd/lib/variants/NAME.so (A/B experiments via OSA_LIB_PATH)
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p openstereo_amd/lib/variants
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops -Iopenstereo_amd/csrc"
for f in openstereo_amd/csrc/conv3d openstereo_amd/csrc/conv_inst_f32 openstereo_amd/csrc/conv_inst_f16x3 openstereo_amd/csrc/conv_inst_f16 openstereo_amd/csrc/conv_march openstereo_amd/csrc/volume openstereo_amd/csrc/wgrad tools/experiments/conv_pipe; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $f.hip -o openstereo_amd/lib/variants/$NAME.$(basename $f).o &
done
wait
OBJS=$(ls openstereo_amd/lib/obj/*.o | grep -v "/conv3d.o\|/conv_inst_f32.o\|/conv_inst_f16x3.o\|/conv_inst_f16.o\|/conv_march.o\|/conv_pipe.o\|/volume.o\|/wgrad.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o openstereo_amd/lib/variants/$NAME.so openstereo_amd/lib/variants/$NAME.*.o $OBJS
mkdir -p openstereo_amd/lib/variants/$NAME.keep
for o in openstereo_amd/lib/variants/$NAME.*.o; do mv $o openstereo_amd/lib/variants/$NAME.keep/$(basename $o | sed "s/^$NAME\.//"); done   # (tools/r4/build_march_variant.sh relinks against them)
echo openstereo_amd/lib/variants/$NAME.so
