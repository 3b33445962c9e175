U3D_LIB_PATH=$PWD/tools/bin/libu3d_nttrace.so python tools/trace_gemm.py 41000 768 256 | grep "slot"
U3D_LIB_PATH=$PWD/tools/bin/libu3d_nttrace_nostag.so python tools/trace_gemm.py 41000 768 256 | grep "slot"
