# PMC passes only (one counter group per pass, MI355X_MICROARCH.md) + the driver's kernel stats, into gpurun_out/<round>/
ROUND=${1:-r04}
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/$ROUND; mkdir -p $O; cd /tmp
rm -rf $O/rp_stats $O/pmc_*
(timeout 200 rocprofv3 --kernel-trace --stats -d $O/rp_stats -o drv --output-format csv -- python $R/tools/profile_driver.py --batch 32 --steps 3) > $O/rp_stats.log 2>&1
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  N=$(echo $P | cut -d" " -f1)
  (timeout 120 rocprofv3 --pmc $P --kernel-trace -d $O/pmc_$N -o drv --output-format csv -- python $R/tools/profile_driver.py --batch 32 --steps 1) > $O/pmc_$N.log 2>&1
done
rm -f $O/rp_*/*trace.csv; du -sh $O
