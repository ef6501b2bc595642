mkdir -p gpurun_out/r04u
for SH in 320,8000,64,21,1.2 320,8000,64,21,1.35 320,8000,64,21,0.5 320,8000,64,21,0.4 320,8000,64,21,1.0; do
  for E in "DPC_NT=5" "DPC_NT=13" "DPC_NT=5" "DPC_NT=13"; do
    echo "== $SH $E" | tee -a gpurun_out/r04u/ab.txt
    env $E AB_SHAPE=$SH timeout 300 python scripts/ab_libs.py differentiable-point-clouds_amd/csrc/libdpc_hip.so 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04u/ab.txt
  done
done
