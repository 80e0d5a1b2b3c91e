# Round-4 GPU call B: the whole GPU suite on the current tree, then A/B of the XCD-group mapping (time + FETCH_SIZE), pp_peaks pruning effect.
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r04b; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests -m gpu -x -q) > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" | tee -a $O/summary.log
for X in 1 0; do
  (timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --engine-opt wino_xcd_groups=$X --dump-profile $O/prof_x$X.json) > $O/bench_x$X.log 2> $O/bench_x$X.err; echo "bench xcd=$X rc=$?" | tee -a $O/summary.log
done
cd /tmp
for X in 1 0; do
  (timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_FETCH_x$X -o drv --output-format csv -- python $R/tools/profile_driver.py --batch 32 --steps 1 --opt wino_xcd_groups=$X) > $O/pmc_FETCH_x$X.log 2>&1
done
cd $R
tail -4 $O/pytest_gpu.log
python - <<'PY'
import json,glob,os,csv,collections
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04b'
for x in (1,0):
    try:
        l=[q for q in open(O+'/bench_x%d.log'%x) if q.startswith('{')][-1]; d=json.loads(l)
        print('xcd_groups',x,'fps %.1f ms %.3f dom avg %.4f ms frac %.3f pp %.3f'%(d['value'],d['ms_per_step'],d['roofline']['avg_launch_ms'],d['roofline']['frac'],d['kernel_time_ms_per_step']['postprocess']))
    except Exception as e: print('bench',x,e)
    try:
        f=glob.glob(O+'/pmc_FETCH_x%d/**/drv_counter_collection.csv'%x,recursive=True)[0]
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r['Counter_Name']=='FETCH_SIZE': agg[r['Kernel_Name'][:60]].append(float(r['Counter_Value']))
        for k,v in agg.items():
            if 'conv_wino_kernel<7, 0, 0, 1>' in k: print('  FETCH_SIZE KiB mean',x,k,sum(v)/len(v),len(v))
    except Exception as e: print('pmc',x,e)
PY
