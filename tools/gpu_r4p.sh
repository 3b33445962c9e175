for r in 64 32; do echo "== R=$r"; U3D_GMM_R=$r PROF_KINDS=wave PROF_MAXLV=2 timeout 120 python tools/prof_gmm.py 10 bf16rows 2>&1 | grep -v "amdgpu.ids\|sum of" | cut -c1-120; done
