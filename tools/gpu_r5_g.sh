# Round-5 GPU call G: where a conv1_wino block's time goes -- diagnostic builds with parts compiled out (results wrong, time only)
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r05g; mkdir -p $O; cd $R
(timeout 900 python tools/kernel_variants.py time --steps 5 --json $O/c1w_ablation.json) 2>&1 | tee $O/c1w_ablation.log
