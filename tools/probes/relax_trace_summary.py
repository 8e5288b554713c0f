#!/usr/bin/env python3
"""rounds / durations per relaxation phase from gpurun_out/relax_trace.txt"""
rows=[l.split() for l in open('gpurun_out/relax_trace.txt')]
phases=[];cur=None
for t,d,nm,g in rows:
    t=float(t);d=float(d)
    if nm in('seed','init','prep'):
        cur={'kind':nm,'relax':[], 'compact':0.0,'t0':t}
        phases.append(cur)
    elif nm=='relax': cur['relax'].append((t,d,int(g)))
    elif nm=='compact': cur['compact']+=d
for p in phases[:2]:
    if not p['relax']: continue
    r=p['relax']
    tot=sum(d for _,d,_ in r)
    print(p['kind'],'rounds',len(r),'relax ms %.1f'%(tot/1e3),'compact ms %.1f'%(p['compact']/1e3),'wall ms %.1f'%((r[-1][0]+r[-1][1]-p['t0'])/1e3))
    ds=[d for _,d,_ in r]
    big=[d for d in ds if d>1000]; mid=[d for d in ds if 100<d<=1000]; small=[d for d in ds if d<=100]
    print('  >1ms: n=%d sum=%.1f ms | 0.1-1ms: n=%d sum=%.1f | <0.1ms: n=%d sum=%.1f'%(len(big),sum(big)/1e3,len(mid),sum(mid)/1e3,len(small),sum(small)/1e3))
    print('  first 40 durations us:',[int(d) for d in ds[:40]])
