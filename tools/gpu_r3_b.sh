# round 3, call B: kernel trace of one batch-32 step (main / tail / combine kernels apart), tail-unit sweep
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r03b; mkdir -p $O; cd /tmp
(timeout 300 rocprofv3 --kernel-trace --stats -d $O/rp -o drv --output-format csv -- python $R/tools/profile_driver.py --batch 32 --steps 3) > $O/rp.log 2>&1
cd $R
python - <<'PY' > $O/trace_summary.txt 2>&1
import csv,glob,collections,re
f=glob.glob('gpurun_out/r03b/rp/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
d=collections.defaultdict(list)
for r in rows:
    n=re.sub(r'\(.*','',r['Kernel_Name'])
    d[(n,r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size',''),r.get('Grid_Size_Z',''))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(d.items(),key=lambda kv:-sum(kv[1])):
    print('%-70s grid %s z %s  n=%d avg %.1f us total %.2f ms'%(k[0][:70],k[1],k[2],len(v),sum(v)/len(v),sum(v)/1e3))
# gaps: last step, time between consecutive kernels
last=rows[-200:]
gaps=[(int(b['Start_Timestamp'])-int(a['End_Timestamp']))/1e3 for a,b in zip(last,last[1:])]
print('gaps between consecutive kernels (us): mean %.2f max %.2f'%(sum(gaps)/len(gaps),max(gaps)))
PY
head -40 $O/trace_summary.txt
for G in 1 2 4; do (timeout 200 python tools/profile_driver.py --batch 32 --steps 5 --opt wino_tail_g=$G) 2>&1 | grep ms/step; done > $O/tail_g.log; cat $O/tail_g.log
