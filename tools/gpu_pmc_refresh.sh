ROUND=r02
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/$ROUND; mkdir -p $O; cd $R
(timeout 300 python tools/profile_driver.py --batch 32 --steps 3 --profile-json $O/prof_strip_tiles.json) > $O/drv_new.log 2>&1
cd /tmp
(timeout 300 rocprofv3 --kernel-trace --stats -d $O/rp_stats -o drv --output-format csv -- python $R/tools/profile_driver.py --batch 32 --steps 3) > $O/rp_stats.log 2>&1
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  N=$(echo $P | cut -d" " -f1)
  (timeout 300 rocprofv3 --pmc $P --kernel-trace -d $O/pmc_$N -o drv --output-format csv -- python $R/tools/profile_driver.py --batch 32 --steps 1) > $O/pmc_$N.log 2>&1
done
cd $R; tail -2 $O/drv_new.log; du -sh $O
