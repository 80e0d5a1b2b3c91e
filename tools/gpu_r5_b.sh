# Round-5 GPU call B: what does a block's store burst cost?  Per-layer times (pmx_conv2d) of diagnostic builds: no output stores, half of
# them, cache-policy bits, a staggered first round -- against the round-4 library and the transposed-tile build; the new RecordPipe test.
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r05b; mkdir -p $O; cd $R
(timeout 900 python tools/kernel_variants.py time-conv --iters 10 --json $O/store_ablation.json) 2>&1 | tee $O/store_ablation.log
(timeout 600 python -m pytest tests/test_gpu_dist.py tests/test_gpu_selection.py -m gpu -x -q) > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.log
tail -5 $O/pytest.log
