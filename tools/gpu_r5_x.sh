# Round-5 GPU call X: unit-mode pieces handed to the XCDs in contiguous ranges (PMX_WINO_UNIT_XCD=1: blocks that share weights share an L2)
# against round-robin (=0): batch 32 (merged tails) and batch 1 (every layer in unit mode), two runs each; then the variant in the product's
# place for the bit-exactness suites
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r05x; mkdir -p $O; cd $R
for i in 1 2; do
(timeout 300 python tools/kernel_variants.py time --steps 5 --json $O/b32_$i.json) 2>&1 | tee $O/b32_$i.log
(timeout 300 python tools/kernel_variants.py time --steps 20 --batch 1 --json $O/b1_$i.json) 2>&1 | tee $O/b1_$i.log
done
cp tools/_build/libpose_var_xcd1.so chainer_realtime_multi-person_pose_estimation_amd/csrc/libpose_mi355x.so
(timeout 900 python -m pytest tests/test_gpu_winograd.py tests/test_gpu_conv.py tests/test_gpu_network.py tests/test_gpu_reference_goldens.py -m gpu -x -q) > $O/pytest.log 2>&1; echo "pytest (xcd1 library) rc=$?" | tee -a $O/summary.log
tail -4 $O/pytest.log
