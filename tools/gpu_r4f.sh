mkdir -p gpurun_out/r4f
for r in 32 64; do for g in 1 3 9; do
echo "== R=$r G=$g" >> gpurun_out/r4f/sweep.txt
U3D_GMM_R=$r U3D_GMM_G=$g PROF_MINLV=3 timeout 120 python tools/prof_gmm.py 10 x3 2>&1 | grep -v "amdgpu.ids\|sum of" | cut -c1-160 >> gpurun_out/r4f/sweep.txt
done; done
cat gpurun_out/r4f/sweep.txt
