mkdir -p gpurun_out/r04cs8
L=differentiable-point-clouds_amd/csrc
DPC_GPU_LIB=$L/libdpc_cs.so timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -p no:cacheprovider \
   -k "goldens or knife or cfg2_full_batch_against or degenerate or test_fused_dropout or fused_candidate_loss or training_shape_at or fused_l2 or asymmetric or fused_path_against or d256 or edge_planes" > gpurun_out/r04cs8/pytest_variant.log 2>&1
echo "pytest exit $?" >> gpurun_out/r04cs8/pytest_variant.log; tail -3 gpurun_out/r04cs8/pytest_variant.log
