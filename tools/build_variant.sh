#!/bin/bash
# A second build of liblara2dgs.so with extra compiler flags, for kernel A/B runs on the GPU box:
#   tools/build_variant.sh <tag> "<extra hipcc flags>"   ->  lara_amd/liblara2dgs_<tag>.so   (use: LARA2DGS_LIB=<that path>)
set -eu
TAG=$1; EXTRA=${2:-}
cd "$(dirname "$0")/../lara_amd/csrc"
OBJ=_obj_$TAG; mkdir -p $OBJ
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function"
for f in abi preprocess binning composite attention encoder encoder_bwd rays surface pointfeat finedec coarsedec loss msssim tsdf; do
  FL=""
  case $f in preprocess|binning|tsdf) FL="-ffp-contract=off";; composite) FL="-fno-slp-vectorize";; esac
  if [ "$f" = composite ] || [ "$f" = encoder_bwd ] || [ "$f" = attention ] || [ "$f" = encoder ] || [ ! -f _obj/$f.o ]; then
    /opt/rocm/bin/hipcc $COMMON $FL $EXTRA -c $f.hip -o $OBJ/$f.o
  else
    cp _obj/$f.o $OBJ/$f.o        # only composite / encoder / encoder_bwd / attention read experiment macros
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../liblara2dgs_$TAG.so $OBJ/*.o
echo built ../liblara2dgs_$TAG.so
