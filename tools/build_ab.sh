#!/bin/bash
# A/B builds of libtg_hip.so: tools/build_ab.sh <tag> <source.hip> [-DNAME=VALUE ...] -> csrc/libtg_<tag>.so, with <source.hip> recompiled
# under the extra flags and every other object taken from the regular build.  Select with TG_LIB=libtg_<tag>.so (tg/lib.py).
set -e
tag=$1; src=$2; shift 2
cd "$(dirname "$0")/../tensorflow-implementation-of-triple-gan_amd/csrc"
make -s libtg_hip.so
base=${src%.hip}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off "$@" -x hip -c $src -o ${base}_ab_${tag}.o 2>&1 | grep -v "warning\|^ *[0-9]* |\|\^\|generated" || true
objs=$(ls *.o | grep -v "_ab_\|_stamp" | grep -v "^${base}.o$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o libtg_${tag}.so ${base}_ab_${tag}.o $objs
echo "built libtg_${tag}.so ($*)"
