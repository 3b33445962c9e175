mkdir -p gpurun_out/r4b
for m in 0 1 2 4 8 16 32 64 128 256 512 7 6; do
  L=tools/bin/libu3d_wgk_$m.so; [ $m = 0 ] && L=unidet3d_amd/csrc/libu3d_hip.so
  echo "== variant $m" >> gpurun_out/r4b/abl.txt
  U3D_LIB_PATH=$PWD/$L PROF_KINDS=workgroup PROF_MAXLV=3 timeout 120 python tools/prof_gmm.py 10 x3 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4b/abl.txt
done
cat gpurun_out/r4b/abl.txt | cut -c1-200
