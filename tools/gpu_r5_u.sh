# Round-5 GPU call U: the clustered transform (PMX_WINO_VCLUSTER) with its first cluster in slot 16 / 20 / 22 against the spread schedule
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r05u; mkdir -p $O; cd $R
(timeout 700 python tools/kernel_variants.py time --steps 5 --json $O/vcl.json) 2>&1 | tee $O/vcl.log
(timeout 700 python tools/kernel_variants.py time --steps 5 --json $O/vcl2.json) 2>&1 | tee $O/vcl2.log
