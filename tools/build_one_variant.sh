#!/bin/bash
# build_one_variant.sh NAME SOURCE(.hip, basename under csrc) [flags]: ONE translation unit compiled with extra flags, linked with the regular objects
set -e
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; shift; shift
mkdir -p openstereo_amd/lib/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops -Iopenstereo_amd/csrc "$@" -c openstereo_amd/csrc/$SRC.hip -o /tmp/$NAME.$SRC.o
OBJS=$(ls openstereo_amd/lib/obj/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o openstereo_amd/lib/variants/$NAME.so /tmp/$NAME.$SRC.o $OBJS
echo openstereo_amd/lib/variants/$NAME.so
