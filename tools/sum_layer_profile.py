import json,sys
d=json.load(open(sys.argv[1])); st=d['steps']
tot=0
rows=[]
for e in d['entries']:
    tot+=e['avg_ms']; rows.append((e['avg_ms'],e['layer'],e['kernel'],e['flop_per_launch']))
print('sum of kernel avg ms: %.3f'%tot)
for ms,l,k,f in sorted(rows,reverse=True)[:int(sys.argv[2]) if len(sys.argv)>2 else 60]:
    print('%8.4f ms %-22s %-34s %6.1f TF/s'%(ms,l,k,f/ms/1e9 if ms else 0))
