# Round-5 GPU call S: second layer of the fused 1x1 pairs on rotating waves (+ padding-only tiles skipped): A/B and bit-exactness
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r05s; mkdir -p $O; cd $R
(timeout 600 python tools/kernel_variants.py time --steps 5 --json $O/pair_rot.json) 2>&1 | tee $O/pair_rot.log
(timeout 600 python tools/kernel_variants.py time --steps 5 --json $O/pair_rot2.json) 2>&1 | tee $O/pair_rot2.log
