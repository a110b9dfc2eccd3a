#!/bin/bash
# Builds a VARIANT of libgpupoly.so next to the shipped one without touching the shipped objects:
#   tools/build_variant.sh phase "PHASE_TIMING=1"    -> mxx_amd/libgpupoly_phase.so
# (sources copied to a scratch dir, `make <args>` there; the .so travels to the GPU box with the snapshot and is selected
# with MXX_GPUPOLY_LIB=mxx_amd/libgpupoly_<name>.so)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
B=/tmp/gpupoly_variant_$NAME
rm -rf $B && mkdir -p $B/mxx_amd/csrc $B/include
cp $ROOT/mxx_amd/csrc/*.hip $ROOT/mxx_amd/csrc/*.h $ROOT/mxx_amd/csrc/*.inc $ROOT/mxx_amd/csrc/Makefile $B/mxx_amd/csrc/
cp $ROOT/include/gpupoly.h $B/include/
make -C $B/mxx_amd/csrc -j8 "$@" > $B/build.log 2>&1 || { grep -E "error" $B/build.log | head; exit 1; }
cp $B/mxx_amd/libgpupoly.so $ROOT/mxx_amd/libgpupoly_$NAME.so
echo "built mxx_amd/libgpupoly_$NAME.so"
