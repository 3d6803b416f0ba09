#!/bin/bash
# Dev helper: variant builds of libgsdf_hip.so that differ in the compile flags of ONE source (compile-time A/B of a kernel), into
# gs-sdf_amd/lib/variants/<name>/libgsdf_hip.so.  usage: tools/build_variants.sh <name> <source.hip> <extra hipcc flags...>  (repeat source/flags pairs with --)
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
NAME=$1; shift
OUT=$ROOT/gs-sdf_amd/lib/variants/$NAME
mkdir -p $OUT/obj
OBJS=$(ls $ROOT/gs-sdf_amd/csrc/_obj/*.o)
while [ $# -gt 0 ]; do
  SRC=$1; shift
  FLAGS=()
  while [ $# -gt 0 ] && [ "$1" != "--" ]; do FLAGS+=("$1"); shift; done
  [ "${1:-}" = "--" ] && shift
  STEM=$(basename $SRC .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$ROOT/gs-sdf_amd/csrc "${FLAGS[@]}" -c $ROOT/gs-sdf_amd/csrc/$SRC -o $OUT/obj/$STEM.o
  OBJS=$(echo "$OBJS" | grep -v "/$STEM.o")
  OBJS="$OBJS $OUT/obj/$STEM.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $OUT/libgsdf_hip.so
echo built $OUT/libgsdf_hip.so
