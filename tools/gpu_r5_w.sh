# Round-5 GPU call W: pass-2 phases with the barrier one step (eight MFMAs) before the phase end (PMX_WINO_P2BAR=1) against the default:
# whole-network layer timing x2, then the variant library in the product's place for the bit-exactness suites
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r05w; mkdir -p $O; cd $R
(timeout 400 python tools/kernel_variants.py time --steps 5 --json $O/p2bar.json) 2>&1 | tee $O/p2bar.log
(timeout 400 python tools/kernel_variants.py time --steps 5 --json $O/p2bar2.json) 2>&1 | tee $O/p2bar2.log
cp tools/_build/libpose_var_p2bar.so chainer_realtime_multi-person_pose_estimation_amd/csrc/libpose_mi355x.so
(timeout 900 python -m pytest tests/test_gpu_winograd.py tests/test_gpu_conv.py tests/test_gpu_network.py tests/test_gpu_reference_goldens.py -m gpu -x -q) > $O/pytest.log 2>&1; echo "pytest (variant library) rc=$?" | tee -a $O/summary.log
tail -4 $O/pytest.log
