#!/bin/bash
# ablation timings of the tile-stationary kernel (variants built by tools/build_variant.sh ... -DU3D_TS_ABL=<mask>)
mkdir -p gpurun_out/r6abl
export PROF_CHECK=0 PROF_MAXLV=${PROF_MAXLV:-1} PROF_SHAPES=${PROF_SHAPES:-32x32} PROF_PLANS=${PROF_PLANS:-128x192}
{
echo "== full kernel"; timeout 300 python tools/prof_ts.py 8 10 2>&1 | grep -v Warn | grep "ts T\|pairs kernel"
for m in "$@"; do
  echo "== U3D_TS_ABL=$m"; U3D_LIB_PATH=tools/bin/libu3d_ts_abl$m.so timeout 300 python tools/prof_ts.py 8 10 2>&1 | grep "ts T"
done
} > gpurun_out/r6abl/abl_${PROF_SHAPES}_${PROF_PLANS}.txt 2>&1
cat gpurun_out/r6abl/abl_${PROF_SHAPES}_${PROF_PLANS}.txt
