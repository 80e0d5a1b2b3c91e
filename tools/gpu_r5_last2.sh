# Round-5 closing run after the clustered transform went to all geometries: whole GPU suite + smoke on the final tree
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r05last2; mkdir -p $O; cd $R
(timeout 290 python -m pytest tests -m gpu -x -q -p no:cacheprovider) > $O/pytest.log 2>&1; echo "pytest gpu rc=$?" | tee -a $O/summary.log
tail -3 $O/pytest.log
(timeout 25 python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.log
