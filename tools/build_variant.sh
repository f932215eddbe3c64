#!/bin/bash
# A/B builds of the SPD pair kernels: tools/build_variant.sh NAME "-DMM_X=1 ..." -> matrix-manifolds_amd/lib/variants/libmm_NAME.so
# (only spd.hip is recompiled; the other objects come from the regular build).  Select at run time with MM_MANIFOLDS_LIB.
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/matrix-manifolds_amd/csrc
mkdir -p $ROOT/matrix-manifolds_amd/build/var_$NAME $ROOT/matrix-manifolds_amd/lib/variants
make -s -C $C
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fno-slp-vectorize --offload-arch=gfx950 -Wno-unused-variable -Wno-unused-but-set-variable "$@" -c $C/spd.hip -o $ROOT/matrix-manifolds_amd/build/var_$NAME/spd.o
OBJS=$(ls $ROOT/matrix-manifolds_amd/build/*.o | grep -v '/spd.o')
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS $ROOT/matrix-manifolds_amd/build/var_$NAME/spd.o -o $ROOT/matrix-manifolds_amd/lib/variants/libmm_$NAME.so
echo built libmm_$NAME.so
