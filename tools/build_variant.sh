#!/bin/bash
# Builds mesh-reconstruction_amd/lib/variants/libmvs_hip_<tag>.so from the current tree with extra compiler flags applied to chosen
# sources (default: csrc/sweep_rect.hip), for A/B timings in ONE GPU session (MVS_HIP_LIBRARY=<that file> python tools/time_general.py ...).
#   tools/build_variant.sh <tag> "<extra flags>" [source ...]
set -e
cd "$(dirname "$0")/../mesh-reconstruction_amd"
tag=$1; extra=$2; shift 2 || true
srcs=${@:-csrc/sweep_rect.hip}
make >/dev/null
tmp=$(mktemp -d)
objs=""
for o in build/*.o; do
    base=$(basename "$o" .o)
    if echo " $srcs " | grep -q " csrc/$base "; then
        x=""; case "$base" in *.cpp) x="-x hip";; esac
        /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -Wall -Wno-unused-function $extra $x -c "csrc/$base" -o "$tmp/$base.o"
        objs="$objs $tmp/$base.o"
    else
        objs="$objs $o"
    fi
done
mkdir -p lib/variants
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "lib/variants/libmvs_hip_$tag.so" $objs -L/opt/rocm/lib -lhipfft -Wl,-rpath,/opt/rocm/lib -ldl -lpthread
rm -rf "$tmp"
echo "lib/variants/libmvs_hip_$tag.so"
