#!/bin/bash
# Build from a SNAPSHOT of the sources: hipcc reads a .hip file twice (device pass, minutes later the host pass), so an
# edit made while a compile is in flight puts two versions of the file into one object (host stubs of kernels the device
# code does not have: hipLaunchKernel aborts in hip::DeviceFunc).  Objects and the library still land in the tree.
#   tools/snap_make.sh                      # the regular library (make -j8)
#   tools/snap_make.sh NAME "-DFLAG ..."    # lib/variants/libmm_NAME.so: spd.hip + spd_loss.hip (or $MM_VARIANT_SRC) recompiled with the flags (MM_SPD_MAX_D=5)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
ID=${1:-main}
S=/tmp/mmsnap/$ID.$$
mkdir -p $S/matrix-manifolds_amd
cp -rp $ROOT/include $S/include
cp -rp $ROOT/matrix-manifolds_amd/csrc $S/matrix-manifolds_amd/csrc
mkdir -p $ROOT/matrix-manifolds_amd/build $ROOT/matrix-manifolds_amd/lib/variants
ln -s $ROOT/matrix-manifolds_amd/build $S/matrix-manifolds_amd/build
ln -s $ROOT/matrix-manifolds_amd/lib $S/matrix-manifolds_amd/lib
if [ "$ID" = main ]; then
  make -j8 -C $S/matrix-manifolds_amd/csrc 2>&1 | grep -E "error|Error|undefined" || true
  ls -la --time-style=full-iso $ROOT/matrix-manifolds_amd/lib/libmm_manifolds.so
else
  shift
  mkdir -p $ROOT/matrix-manifolds_amd/build/var_$ID
  OBJS=$(ls $ROOT/matrix-manifolds_amd/build/*.o)
  : > /tmp/build_$ID.log
  PIDS=""
  for SRC in ${MM_VARIANT_SRC:-spd.hip spd_loss.hip}; do   # (the translation units that include spd_pair.hpp / smallmat.hpp's SPD paths)
    OBJ=${SRC%.hip}.o
    OBJS=$(echo "$OBJS" | grep -v "/$OBJ\$")
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fno-slp-vectorize --offload-arch=gfx950 -Wno-unused-variable -Wno-unused-but-set-variable \
      -DMM_SPD_MAX_D=5 -Rpass-analysis=kernel-resource-usage "$@" -c $S/matrix-manifolds_amd/csrc/$SRC \
      -o $ROOT/matrix-manifolds_amd/build/var_$ID/$OBJ > /tmp/build_${ID}_${SRC%.hip}.log 2>&1 &
    PIDS="$PIDS $!"
    OBJS="$OBJS $ROOT/matrix-manifolds_amd/build/var_$ID/$OBJ"
  done
  for P in $PIDS; do wait $P; done
  cat /tmp/build_${ID}_*.log > /tmp/build_$ID.log
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS -ldl -o $ROOT/matrix-manifolds_amd/lib/variants/libmm_$ID.so
  echo built libmm_$ID.so
fi
rm -rf $S
