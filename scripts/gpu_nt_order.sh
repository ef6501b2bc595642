mkdir -p gpurun_out/r04cs6
L=differentiable-point-clouds_amd/csrc
for SH in 32,8000,128,11,1.6 8,16000,256,11,2.0 320,8000,64,21,3.0 320,8000,64,21,0.8 320,8000,64,21,0.3; do
  echo "== $SH" | tee -a gpurun_out/r04cs6/ab.txt
  AB_SHAPE=$SH timeout 300 python scripts/ab_libs.py $L/libdpc_hip.so $L/libdpc_cs.so 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04cs6/ab.txt
done
