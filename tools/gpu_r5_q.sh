# Round-5 GPU call Q: which two tiles share a 16-lane LDS access in the Winograd input transform (neighbours vs four tiles apart): A/B
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r05q; mkdir -p $O; cd $R
(timeout 600 python tools/kernel_variants.py time --steps 5 --json $O/tperm.json) 2>&1 | tee $O/tperm.log
(timeout 600 python tools/kernel_variants.py time --steps 5 --json $O/tperm2.json) 2>&1 | tee $O/tperm2.log
