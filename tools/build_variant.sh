#!/bin/bash
# build_variant.sh NAME [extra hipcc flags]: conv3d.hip compiled with extra flags, linked with the
# regular objects into openstereo_amd/lib/variants/NAME.so (A/B experiments via OSA_LIB_PATH)
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p openstereo_amd/lib/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off "$@" -c openstereo_amd/csrc/conv3d.hip -o openstereo_amd/lib/variants/$NAME.o
OBJS=$(ls openstereo_amd/lib/obj/*.o | grep -v conv3d.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o openstereo_amd/lib/variants/$NAME.so openstereo_amd/lib/variants/$NAME.o $OBJS
rm openstereo_amd/lib/variants/$NAME.o
echo openstereo_amd/lib/variants/$NAME.so
