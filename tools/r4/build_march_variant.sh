#!/bin/bash
# build_march_variant.sh NAME [flags]: conv_march.hip alone with extra flags (seconds), linked with the -DOSA_EXPERIMENTS objects of
# tools/build_variant.sh exp (which must exist: openstereo_amd/lib/variants/exp.keep/*.o) -> openstereo_amd/lib/variants/NAME.so
set -e
cd "$(dirname "$0")/../.."
NAME=$1; shift
V=openstereo_amd/lib/variants
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -Iopenstereo_amd/csrc -DOSA_EXPERIMENTS"
/opt/rocm/bin/hipcc $FLAGS "$@" -c openstereo_amd/csrc/conv_march.hip -o $V/$NAME.conv_march.o
OBJS=$(ls openstereo_amd/lib/obj/*.o | grep -v "/conv3d.o\|/conv_inst_f32.o\|/conv_inst_f16x3.o\|/conv_inst_f16.o\|/conv_march.o\|/conv_pipe.o\|/volume.o\|/wgrad.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/$NAME.so $V/$NAME.conv_march.o $V/exp.keep/conv3d.o $V/exp.keep/conv_inst_f32.o $V/exp.keep/conv_inst_f16x3.o $V/exp.keep/conv_inst_f16.o $V/exp.keep/volume.o $V/exp.keep/wgrad.o $V/exp.keep/conv_pipe.o $OBJS
rm $V/$NAME.conv_march.o
echo $V/$NAME.so
