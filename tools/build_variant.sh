#!/bin/bash
# usage: tools/build_variant.sh <commit|WORK> <tag>  -> ra-slam_amd/csrc/build/libratsdf_<tag>.so (same C ABI; select with RATSDF_LIB)
# Same-box A/B of engine variants: bash tools/ab_env.sh N "<bench args>" RATSDF_LIB=$PWD/ra-slam_amd/csrc/build/libratsdf_a.so RATSDF_LIB=...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
c=$1; tag=$2
T=$(mktemp -d)
if [ "$c" = WORK ]; then cp -r $ROOT/ra-slam_amd/csrc $T/csrc; mkdir -p $T/include; cp $ROOT/include/ratsdf.h $T/include/; mv $T/csrc $T/x; mkdir -p $T/ra-slam_amd; mv $T/x $T/ra-slam_amd/csrc
else (cd $ROOT && git archive $c ra-slam_amd/csrc include) | tar -x -C $T; fi
rm -rf $T/ra-slam_amd/csrc/build
make -C $T/ra-slam_amd/csrc > $T/log 2>&1 || { tail $T/log; exit 1; }
cp $T/ra-slam_amd/csrc/build/libratsdf.so $ROOT/ra-slam_amd/csrc/build/libratsdf_$tag.so
rm -rf $T
echo "built libratsdf_$tag.so from $c"
