mkdir -p gpurun_out/r4n
for r in 32 64; do for g in 1 3 9; do
echo "== R=$r G=$g" >> gpurun_out/r4n/sweep_bf16rows.txt
U3D_GMM_R=$r U3D_GMM_G=$g PROF_KINDS=wave PROF_MINLV=3 timeout 120 python tools/prof_gmm.py 10 bf16rows 2>&1 | grep -v "amdgpu.ids\|sum of" | cut -c1-120 >> gpurun_out/r4n/sweep_bf16rows.txt
done; done
cat gpurun_out/r4n/sweep_bf16rows.txt
