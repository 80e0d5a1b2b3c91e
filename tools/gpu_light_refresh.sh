# Light refresh of the round artefacts whose inputs changed without a kernel change (selection rule, host code): bench.py, the same
# command under rocprofv3, the batch-1 kernel trace, the batch sweep.  Usage: gpurun -- 'bash tools/gpu_light_refresh.sh r02'
ROUND=${1:-r01}
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/$ROUND; mkdir -p $O; cd $R
(timeout 600 python bench.py --steps 5 --warmup 2 --dump-profile $O/prof_bench.json) > $O/bench.log 2>&1
cd /tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d $O/rp_bench -o bench --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline) > $O/rp_bench.log 2>&1
(timeout 300 rocprofv3 --kernel-trace --stats -d $O/rp_b1 -o b1 --output-format csv -- python $R/tools/profile_driver.py --batch 1 --steps 20) > $O/rp_b1.log 2>&1
cd $R
(timeout 300 python tools/wino_batch_sweep.py) > $O/wino_batch_sweep.txt 2>&1
tail -1 $O/bench.log | cut -c1-400; tail -3 $O/wino_batch_sweep.txt
