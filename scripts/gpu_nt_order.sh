mkdir -p gpurun_out/r04cs5
L=differentiable-point-clouds_amd/csrc
DPC_GPU_LIB=$L/libdpc_cs.so timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -p no:cacheprovider \
   -k "goldens or knife or cfg2_full_batch_against or cfg5_full or degenerate or test_fused_dropout or fused_candidate_loss or training_shape_at or fused_l2 or asymmetric or fused_path_against or d256 or edge_planes" > gpurun_out/r04cs5/pytest_variant.log 2>&1
echo "pytest exit $?" >> gpurun_out/r04cs5/pytest_variant.log; tail -3 gpurun_out/r04cs5/pytest_variant.log
for SH in 8,16000,256,11,2.0 320,8000,64,21,0.8 320,8000,64,21,1.2; do
  echo "== $SH" | tee -a gpurun_out/r04cs5/ab.txt
  AB_SHAPE=$SH timeout 300 python scripts/ab_libs.py $L/libdpc_hip.so $L/libdpc_cs.so 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04cs5/ab.txt
done
