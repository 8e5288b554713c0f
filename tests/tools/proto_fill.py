"""numpy model of the GPU fill algorithm (descent forest -> basins -> Boruvka rounds).
pair_list=False: every round re-reads the raster (the r01d engine, what RDGPU_FILL_EDGES=0 runs);
pair_list=True:  the raster is read once -- every adjacent cell pair is looked at by its earlier cell (E, SE, S, SW
                 neighbours; D4: E, S), the lowest pass per component pair is kept, and rounds 2.. contract that
                 list (k_scan<EMIT> + k_edge_round).
Algorithm validation only; not product, not oracle."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle
from richdem_amd.synth import fractal_dem, fractal_dem_int

SH8=[(0,-1),(-1,-1),(-1,0),(-1,1),(0,1),(1,1),(1,0),(1,-1)]  # (dy,dx) n=1..8
SH4=[(0,-1),(-1,0),(0,1),(1,0)]

def shifted(a, dy, dx, fill):
    h,w=a.shape
    out=np.full_like(a, fill)
    ys=slice(max(0,-dy), min(h,h-dy)); xs=slice(max(0,-dx), min(w,w-dx))
    ysn=slice(max(0,dy), min(h,h+dy)); xsn=slice(max(0,dx), min(w,w+dx))
    out[ys,xs]=a[ysn,xsn]
    return out

def fill_proto(z, topo=8, roots_mask=None, verbose=False, pair_list=False):
    SH=SH8 if topo==8 else SH4
    h,w=z.shape; N=h*w
    # order preserving integer key
    ranks=np.unique(z, return_inverse=True)[1].reshape(h,w).astype(np.int64)+1  # >=1
    k=ranks
    idx=np.arange(N,dtype=np.int64).reshape(h,w)
    OUT=np.int64(N)
    border=np.zeros((h,w),bool); border[0,:]=border[-1,:]=border[:,0]=border[:,-1]=True
    # phase 0
    bestk=k.copy(); besti=idx.copy()
    BIG=np.int64(1<<60)
    for dy,dx in SH:
        kn=shifted(k,dy,dx,BIG); inn=shifted(idx,dy,dx,BIG)
        better=(kn<bestk)|((kn==bestk)&(inn<besti))
        bestk=np.where(better,kn,bestk); besti=np.where(better,inn,besti)
    ptr=besti.copy()
    ptr[border]=OUT
    ptr=ptr.ravel()
    # phase 1 jump
    ext=np.append(ptr,OUT)
    it=0
    while True:
        nxt=ext[ext]
        it+=1
        if (nxt==ext).all(): break
        ext=nxt
    ptr=ext[:N]
    pits=np.flatnonzero(ptr==np.arange(N))
    B=len(pits)
    pitid=np.full(N+1,-1,np.int64); pitid[pits]=np.arange(B); pitid[N]=B   # OUT -> B
    lab=pitid[ptr].reshape(h,w)
    assert (lab>=0).all()
    OUTC=B
    cur=np.arange(B+1); acc=np.zeros(B+1,np.int64)
    kflat=k
    rounds=0
    pairs=None   # pair list: arrays (a, b, key), a < b, one record per adjacent component pair
    npairs0=0
    while True:
        comp=cur[lab]
        if (cur==OUTC).all(): break
        rounds+=1
        best=np.full(B+1,np.iinfo(np.int64).max,np.int64)
        if pair_list and pairs is None:
            # the one raster pass: forward neighbours only, so every adjacent cell pair is seen exactly once
            FWD=[(0,1),(1,1),(1,0),(1,-1)] if topo==8 else [(0,1),(1,0)]   # (dy,dx): E, SE, S, SW
            pa=[];pb=[];pk=[]
            for dy,dx in FWD:
                kn=shifted(kflat,-dy,-dx,-1); cn=shifted(comp,-dy,-dx,-1)   # value AT the neighbour (y+dy, x+dx)
                m=(cn>=0)&(cn!=comp)
                pa.append(np.minimum(comp,cn)[m]); pb.append(np.maximum(comp,cn)[m]); pk.append(np.maximum(kflat,kn)[m])
            pa=np.concatenate(pa);pb=np.concatenate(pb);pk=np.concatenate(pk)
            code=pa*(B+2)+pb
            order=np.lexsort((pk,code)); code=code[order]; pk=pk[order]
            first=np.r_[True,code[1:]!=code[:-1]]           # lowest pass per pair
            pairs=(code[first]//(B+2), code[first]%(B+2), pk[first]); npairs0=len(pairs[0])
        if pair_list:
            a,b,kk=pairs
            ca=cur[a]; cb=cur[b]
            live=ca!=cb                                      # the only closed component is the outside: never both
            ca,cb,kk=ca[live],cb[live],kk[live]
            lo=np.minimum(ca,cb); hi=np.maximum(ca,cb)
            code=lo*(B+2)+hi
            order=np.lexsort((kk,code)); code=code[order]; kk=kk[order]
            first=np.r_[True,code[1:]!=code[:-1]] if len(code) else np.zeros(0,bool)
            lo=code[first]//(B+2); hi=code[first]%(B+2); kk=kk[first]
            pairs=(lo,hi,kk)                                 # merged per pair: next round's list
            mo=lo!=OUTC; np.minimum.at(best,lo[mo],kk[mo]*(B+2)+hi[mo])
            mo=hi!=OUTC; np.minimum.at(best,hi[mo],kk[mo]*(B+2)+lo[mo])
        else:
            # candidates
            cw=[];cc=[];ct=[]
            for dy,dx in SH:
                kn=shifted(kflat,dy,dx,-1); cn=shifted(comp,dy,dx,-1)
                m=(cn>=0)&(cn!=comp)&(comp!=OUTC)
                cw.append(np.maximum(kflat,kn)[m]); cc.append(comp[m]); ct.append(cn[m])
            cw=np.concatenate(cw);cc=np.concatenate(cc);ct=np.concatenate(ct)
            key=cw*(B+2)+ct
            np.minimum.at(best,cc,key)
        roots=np.flatnonzero((cur==np.arange(B+1))&(np.arange(B+1)!=OUTC))
        bw=best[roots]//(B+2); bt=best[roots]%(B+2)
        assert (best[roots]<np.iinfo(np.int64).max).all()
        par=np.arange(B+1); pm=np.zeros(B+1,np.int64)
        par[roots]=bt; pm[roots]=bw
        # mutual pairs: smaller id stays root
        mutual=(par[par[roots]]==roots)&(bt!=OUTC)&(roots<bt)
        par[roots[mutual]]=roots[mutual]; pm[roots[mutual]]=0
        # pointer jumping with max
        while True:
            pp=par[par]; mm=np.maximum(pm,pm[par])
            if (pp==par).all(): break
            par=pp; pm=mm
        # update basins
        c=cur.copy()
        cur=par[c]; acc=np.maximum(acc,pm[c])
        if verbose: print('round',rounds,'roots',len(roots),'->',int(((cur==np.arange(B+1))).sum()-1))
    # final
    L=acc[lab]
    kk=np.maximum(k,L)
    # map rank back to value
    vals=np.unique(z)
    out=vals[kk-1]
    return out, dict(jump_iters=it, basins=B, rounds=rounds, pair_records=npairs0)

if __name__=='__main__':
    oracle.build()
    for (w,h,seed) in [(200,150,1),(513,257,2),(64,64,3)]:
        for mk in ('f','i','i2'):
            z = fractal_dem(w,h,seed) if mk=='f' else fractal_dem_int(w,h,seed, 1.0 if mk=='i' else 0.05)
            for topo in (8,4):
                ref=oracle.port.fill(z,topo)
                for pl in (False,True):
                    got,info=fill_proto(z,topo,pair_list=pl)
                    print(w,h,seed,mk,topo,pl,(got==ref).all(),info, 'changed',(ref!=z).mean())
                    assert (got==ref).all()
