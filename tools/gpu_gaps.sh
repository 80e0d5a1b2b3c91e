R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r03gap; mkdir -p $O; cd /tmp
(timeout 300 rocprofv3 --kernel-trace -d $O/rp -o drv --output-format csv -- python $R/tools/profile_driver.py --batch 32 --steps 4) > $O/rp.log 2>&1
cd $R
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r03gap/rp/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last step: find last prep_u8 kernel
idx=[i for i,r in enumerate(rows) if 'prep_u8' in r['Kernel_Name']]
s=idx[-1]; e=len(rows)
# end at last pp_group
pg=[i for i,r in enumerate(rows) if 'pp_group' in r['Kernel_Name']]
e=pg[-1]+1
step=rows[s:e]
span=(int(step[-1]['End_Timestamp'])-int(step[0]['Start_Timestamp']))/1e3
busy=sum((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in step)
gaps=[(int(b['Start_Timestamp'])-int(a['End_Timestamp']))/1e3 for a,b in zip(step,step[1:])]
print('kernels',len(step),'span us',span,'busy us',busy,'gap total',sum(gaps),'mean gap',sum(gaps)/len(gaps),'max',max(gaps))
big=sorted(((g,step[i]['Kernel_Name'][:50],step[i+1]['Kernel_Name'][:50]) for i,g in enumerate(gaps)),reverse=True)[:8]
for b in big: print(b)
# previous step end to this step start
prev=rows[s-1]
print('inter-step gap us',(int(step[0]['Start_Timestamp'])-int(prev['End_Timestamp']))/1e3, prev['Kernel_Name'][:40])
PY
rm -rf $O/rp
