mkdir -p gpurun_out/r04cs2
CS=differentiable-point-clouds_amd/csrc/libdpc_cs.so
for SH in 32,8000,128,11,1.6 8,16000,256,11,2.0 320,8000,64,21,3.0 320,8000,64,21,0.8 320,8000,64,21,0.3; do
  echo "== $SH" | tee -a gpurun_out/r04cs2/ab.txt
  AB_SHAPE=$SH timeout 300 python scripts/ab_libs.py differentiable-point-clouds_amd/csrc/libdpc_hip.so $CS 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04cs2/ab.txt
done
