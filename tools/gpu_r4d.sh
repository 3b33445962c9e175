mkdir -p gpurun_out/r4d
for r in 64 32; do echo "== U3D_GMM_R=$r" >> gpurun_out/r4d/r.txt; U3D_GMM_R=$r PROF_MAXLV=3 timeout 200 python tools/prof_gmm.py 10 x3 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4d/r.txt; done
cat gpurun_out/r4d/r.txt | cut -c1-220
