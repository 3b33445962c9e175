#!/bin/bash
# Build a variant of the kernel library for A/B runs (U3D_LIB_PATH=<out> python bench.py ...):
#   tools/build_variant.sh <out.so> <file.hip[,file2.hip...]> <extra hipcc flags...>
#   e.g.  tools/build_variant.sh tools/bin/libu3d_ald40.so spconv.hip -DU3D_GMM_ALD40
# The named sources are recompiled with the extra flags, every other object comes from the in-tree build.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); OUT=$1; SRCS=${2//,/ }; shift 2
C=$R/unidet3d_amd/csrc; TMP=$(mktemp -d)
for SRC in $SRCS; do
    EXTRA=""; { [ "$SRC" = "spconv.hip" ] || [ "$SRC" = "spconv_wg.hip" ] || [ "$SRC" = "attn_x3.hip" ]; } && EXTRA="-mllvm -amdgpu-mfma-vgpr-form"; [ "$SRC" = "postproc.hip" ] && EXTRA="-ffp-contract=off"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I $R/include -I $C $EXTRA "$@" -c $C/$SRC -o $TMP/${SRC%.hip}.o &
done
wait
for SRC in $SRCS; do [ -f $TMP/${SRC%.hip}.o ] || { echo "compile of $SRC failed"; exit 1; }; done
OBJS=""; for f in $C/*.o; do b=$(basename $f); [ -f $TMP/$b ] && OBJS="$OBJS $TMP/$b" || OBJS="$OBJS $f"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS
rm -rf $TMP; ls -la $OUT
