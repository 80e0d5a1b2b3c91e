# round 3, call A: run-geometry Winograd kernel -- parity tests of the new launch forms, then A/B timing at batch 32
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r03a; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests/test_gpu_winograd.py -q -x -p no:cacheprovider -k "run_geometry or run_tail or runs_and_tails or c_twin or selection" 2>&1 | tail -30) > $O/pytest_wino.log
tail -5 $O/pytest_wino.log
for OPTS in "--opt wino_geom=0" "--opt wino_tail=0" ""; do
  N=$(echo "$OPTS" | tr -d ' =-' ); N=${N:-default}
  (timeout 300 python tools/profile_driver.py --batch 32 --steps 5 $OPTS) > $O/drv_$N.log 2>&1; tail -2 $O/drv_$N.log | head -1
  (timeout 300 python tools/profile_driver.py --batch 32 --steps 3 $OPTS --profile-json $O/prof_$N.json) > $O/drvp_$N.log 2>&1
done
for B in 8 16; do (timeout 200 python tools/profile_driver.py --batch $B --steps 5 --opt wino_geom=0) 2>&1 | grep ms/step; (timeout 200 python tools/profile_driver.py --batch $B --steps 5) 2>&1 | grep ms/step; done > $O/drv_small.log
cat $O/drv_small.log
python tools/sum_layer_profile.py $O/prof_default.json 2>/dev/null | head -60
