#!/bin/bash
# build_head_variant.sh NAME [flags]: softargmin.hip alone compiled with extra flags, linked with the regular objects -> lib/variants/NAME.so
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p openstereo_amd/lib/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -Iopenstereo_amd/csrc "$@" -c openstereo_amd/csrc/softargmin.hip -o /tmp/$NAME.softargmin.o
OBJS=$(ls openstereo_amd/lib/obj/*.o | grep -v "/softargmin.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o openstereo_amd/lib/variants/$NAME.so /tmp/$NAME.softargmin.o $OBJS
echo openstereo_amd/lib/variants/$NAME.so
