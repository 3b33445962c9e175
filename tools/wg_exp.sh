#!/bin/bash
# sweep of the weight-gradient row-tile count (U3D_WGRAD_TILES) per level / batch / operand type with tools/prof_wgrad.py;
# output kept in profiles/round2_wgrad_tile_sweep.txt.   usage (GPU box): bash tools/wg_exp.sh
cd $GRAFT_REPO_ROOT
for B in 8 16; do for lvl in 1 2; do for cap in 128 256 512 1024 2048; do for m in fp32 bf16; do
U3D_WGRAD_TILES=$cap U3D_WGRAD_PARTIAL_MB=512 python tools/prof_wgrad.py $B $lvl $m 2>/dev/null | tail -1
done; done; done; done
