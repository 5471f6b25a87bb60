#!/bin/bash
# usage: scripts/dev/build_variant.sh <name> <file.hip> "<extra flags>"  -> tabmat_amd/_abl/libtabmat_<name>.so
# (one source file recompiled with extra -D flags, linked against the other objects of the regular build)
set -e
cd /root/repo/tabmat_amd/csrc
mkdir -p ../_abl/_o_$1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function $3 -c $2 -o ../_abl/_o_$1/${2%.hip}.o
objs=$(ls _build/*.o | grep -v "_build/${2%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../_abl/libtabmat_$1.so $objs ../_abl/_o_$1/${2%.hip}.o
echo built ../_abl/libtabmat_$1.so
