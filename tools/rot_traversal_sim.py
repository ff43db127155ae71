"""NumPy replay of the rotate adjoint traversal (tile catchments, exact row intervals, 4 lanes per row, 16 rows per wave) at\nthe headline shape: lane use of the row order as it is, of other orders, and of an ideal sort by trip count."""
import numpy as np, sys
sys.path.insert(0,'/root/repo')
from neural_flow_style_amd import synthetic as S
G=200; TZ,TY,TX=14,14,34; GROUP=4; NGRP=1024//GROUP
mats=S.uniform_views(8)
n=np.array([G,G,G],float)
def tile_stats(z0,y0,x0,z1,y1,x1):
    tot_samples=0; tot_iter_cur=0; tot_iter_sorted=0; tot_rows=0; empty=0
    for R in mats:
        R=np.asarray(R,float)
        A=np.zeros((3,3)); c=np.zeros(3)
        ha=0.5*(n-1)
        for a in range(3):
            for b in range(3):
                A[a,b]=R[a,b]*(2.0/(n[b]-1))*ha[a]
            c[a]=(1.0-R[a].sum())*ha[a]
        tlo=np.array([z0,y0,x0],float); thi=np.array([z1,y1,x1],float)
        xl=tlo-1.05; xh=thi+1.05
        for a in range(3):
            mn=np.minimum(A[a]*(n-1),0).sum(); mx=np.maximum(A[a]*(n-1),0).sum()
            if tlo[a]==0: xl[a]=min(c[a]+mn-0.5,-1.05)
            if thi[a]==n[a]-1: xh[a]=max(c[a]+mx+0.5,n[a]+0.05)
        inv=np.linalg.inv(A)
        mid=0.5*(xl+xh)-c; half=0.5*(xh-xl)
        oc=inv@mid; oe=np.abs(inv)@half
        lo=np.clip(np.floor(oc-oe),0,n).astype(int); hi=np.clip(np.ceil(oc+oe),-1,n-1).astype(int)
        if hi[2]<lo[2]: continue
        oz,oy=np.meshgrid(np.arange(lo[0],hi[0]+1),np.arange(lo[1],hi[1]+1),indexing='ij')
        oz=oz.ravel().astype(float); oy=oy.ravel().astype(float)
        ta=np.full(oz.shape,float(lo[2])); tb=np.full(oz.shape,float(hi[2]))
        for a in range(3):
            p=c[a]+A[a,0]*oz+A[a,1]*oy
            s=A[a,2]
            if abs(s)>1e-6:
                u0=(xl[a]-p)/s; u1=(xh[a]-p)/s
                ta=np.maximum(ta,np.minimum(u0,u1)-0.01); tb=np.minimum(tb,np.maximum(u0,u1)+0.01)
            else:
                bad=(p<xl[a])|(p>xh[a]); tb=np.where(bad,ta-2,tb)
        xa=np.ceil(ta); xb=np.floor(tb)
        ln=np.maximum(xb-xa+1,0).astype(int)
        its=-(-ln//GROUP)
        tot_samples+=ln.sum(); tot_rows+=len(ln); empty+=(ln==0).sum()
        # current order: rows assigned grp, grp+NGRP..; wave = 16 consecutive groups
        R_=len(ln); pad=(-R_)%NGRP
        it_p=np.concatenate([its,np.zeros(pad,int)]).reshape(-1,NGRP)   # passes x groups
        per_group=it_p   # each pass: wave max over 16 groups
        wave_max=it_p.reshape(it_p.shape[0],NGRP//16,16).max(2)
        tot_iter_cur+=wave_max.sum()*16
        ez=hi[0]-lo[0]+1; ey=hi[1]-lo[1]+1
        itz=its.reshape(ez,ey).T.ravel()   # z fastest
        padz=(-len(itz))%NGRP
        wz=np.concatenate([itz,np.zeros(padz,int)]).reshape(-1,NGRP//16,16).max(2)
        global ZF
        ZF+=wz.sum()*16
        global PF, SL
        # pair fold: group handles rows i and R-1-i back to back
        Rn=len(its); half=(Rn+1)//2
        pair=its[:half].copy(); pair[:Rn-half]+=its[::-1][:Rn-half]
        pp=(-len(pair))%16
        PF+=np.concatenate([pair,np.zeros(pp,int)]).reshape(-1,16).max(1).sum()*16
        # fold within each z slab: rows of slab sorted by |oy - centre| pairing (oy fastest order, pair j with ey-1-j)
        sl=its.reshape(ez,ey)
        h2=(ey+1)//2
        ps=sl[:,:h2].copy(); ps[:,:ey-h2]+=sl[:,::-1][:,:ey-h2]
        psr=ps.ravel(); pp=(-len(psr))%16
        SL+=np.concatenate([psr,np.zeros(pp,int)]).reshape(-1,16).max(1).sum()*16
        srt=np.sort(its)[::-1]; srt=srt[srt>0]
        pad=(-len(srt))%16
        sw=np.concatenate([srt,np.zeros(pad,int)]).reshape(-1,16).max(1)
        tot_iter_sorted+=sw.sum()*16
    return tot_samples,tot_iter_cur,tot_iter_sorted,tot_rows,empty
ZF=0; PF=0; SL=0
acc=np.zeros(5)
tz=(G+TZ-1)//TZ; ty=(G+TY-1)//TY; tx=(G+TX-1)//TX
import itertools
for bz,by,bx in itertools.product(range(0,tz,3),range(0,ty,3),range(tx)):
    z0,y0,x0=bz*TZ,by*TY,bx*TX
    acc+=np.array(tile_stats(z0,y0,x0,min(z0+TZ,G)-1,min(y0+TY,G)-1,min(x0+TX,G)-1))
s,cur,srt,rows,empty=acc
print("pair-fold lane use %.2f, slab-fold %.2f" % (acc[0]/(PF*GROUP), acc[0]/(SL*GROUP)))
print("z-fastest lane use %.2f" % (acc[0]/(ZF*GROUP)))
print("samples %.0f; group-iterations x4 lanes: current %.0f (lane use %.2f), sorted by length %.0f (lane use %.2f); rows %d empty %d" % (s,cur*GROUP,s/(cur*GROUP),srt*GROUP,s/(srt*GROUP),rows,empty))
