mkdir -p gpurun_out/r4micro
{ echo "== tools/coissue_bf16.hip (VALU next to bf16 MFMAs, same wave and 1-4 waves per SIMD)"; timeout 60 tools/bin/coissue_bf16; echo; echo "== tools/l1_bw.hip (bytes per clock and CU of 16-byte loads by access shape and residency)"; timeout 60 tools/bin/l1_bw; echo; echo "== ds_read_b64_tr_b16 probe: lds[i] = i, lane t of a 16-lane group points at elements 4t..4t+3 of its group's 64; each lane prints what it received"; timeout 60 tools/bin/tr_probe | head -20; } > gpurun_out/r4micro/microbench.txt 2>&1
tail -5 gpurun_out/r4micro/microbench.txt
