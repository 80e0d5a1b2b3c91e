R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/pmc_conv; rm -rf $O; mkdir -p $O; cd /tmp
for CFG in "18 0" "18 83968" "10 0"; do
  set -- $CFG; V=$1; L=$2
  for P in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_LDS_BANK_CONFLICT"; do
    N=$(echo $P | cut -d" " -f1)
    (timeout 200 rocprofv3 --pmc $P --kernel-trace -d $O/v${V}_l${L}_$N -o c --output-format csv -- python $R/tools/conv_one.py --variant $V --min-lds $L --iters 2) > $O/v${V}_l${L}_$N.log 2>&1
  done
done
cd $R; python - <<'PY'
import csv, glob, os, collections, statistics
O=os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out','pmc_conv')
for d in sorted(glob.glob(O+'/*/c_counter_collection.csv')):
    agg=collections.defaultdict(list); dur=[]
    for r in csv.DictReader(open(d)):
        if 'conv_mfma' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value'])); dur.append(float(r['End_Timestamp'])-float(r['Start_Timestamp']))
    print(os.path.basename(os.path.dirname(d)), 'dur_us %.1f' % (statistics.mean(dur)/1e3 if dur else 0), {k: '%.4g' % statistics.mean(v) for k,v in agg.items()})
PY
