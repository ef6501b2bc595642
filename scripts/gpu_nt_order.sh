# bench.py / kernel-level A/B of run-time switches of the shipped library (edit the lists): DPC_VIEW_ORDER, DPC_NT, DPC_NT_STORES
mkdir -p gpurun_out/r04ad
for SH in 320,8000,64,21,0.3 320,8000,64,21,0.4 320,8000,64,21,0.5 32,8000,128,5,0.8 4,1000,64,3,0.4; do
  echo "== $SH" | tee -a gpurun_out/r04ad/ab.txt
  AB_SHAPE=$SH timeout 300 python scripts/ab_libs.py differentiable-point-clouds_amd/csrc/libdpc_hip.so differentiable-point-clouds_amd/csrc/libdpc_f12.so differentiable-point-clouds_amd/csrc/libdpc_f18.so differentiable-point-clouds_amd/csrc/libdpc_f24.so 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04ad/ab.txt
done
