"""LDS bank conflicts of compute_tau_absorption's gathers, simulated on the bench atmosphere (CPU, oracle interpolation): every ds_read_b128 of
the major gather (8 stencil rows per column) and of a minor gather (4 rows), the hardware's four 16-lane groups, one 4-bank window per distinct
row -- cycles per group access under the current slab layout (row stride 18 doubles) and under alternatives (pair index rotated / XOR-ed by row
bits).  DESIGN.md section 4.2c.  usage: python tools/lds_conflict_sim.py"""
import sys, numpy as np, itertools
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from rte_rrtmgp_amd import frontend, synth
from oracle import oracle as O
lib=O.load_c(); xp=frontend.NumpyArrays()
ncol,nlay=2048,60
kd=synth.make_kdist("lw"); atm=synth.make_atmosphere(ncol,nlay,seed=42,kdist=kd)
go=frontend.GasOptics(lib,kd,xp)
st=go.interpolation(ncol,nlay,atm.play,atm.tlay,atm.col_gas)
jT=np.asarray(st.jtemp); jp=np.asarray(st.jpress); tropo=np.asarray(st.tropo).astype(bool); jeta=np.asarray(st.jeta)
gf=np.asarray(kd.arrays["gpoint_flavor"]); bl=np.asarray(kd.arrays["band_lims_gpt"])
groups=[[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
groups+= [[l+32 for l in g] for g in groups]
samples=[]  # per group access: arrays (row, p, t, e) of distinct rows, plus nT,nE
for t0 in range(0,ncol,512):
  for l in range(0,nlay,5):
    for b in range(16):
      gS=bl[0,b]-1; cols=slice(t0,t0+512)
      itr=(~tropo[cols,l]).astype(int)
      fl=np.where(itr==0, gf[0,gS], gf[1,gS])-1
      T=jT[cols,l]; P=jp[cols,l]+itr+1
      e1=jeta[0,cols,l,:][np.arange(512),fl]; e2=jeta[1,cols,l,:][np.arange(512),fl]
      Tmin=T.min(); Pmin=P.min()-1; emin=min(e1.min(),e2.min()); emax=max(e1.max(),e2.max())+1
      nT=T.max()+1-Tmin+1; nE=emax-emin+1
      for w in range(8):
        ln=slice(64*w,64*w+64)
        for dp,dt,de,ee in ((0,0,0,e1),(0,0,1,e1),(1,0,0,e1),(1,0,1,e1),(0,1,0,e2),(0,1,1,e2),(1,1,0,e2),(1,1,1,e2)):
          p_=(P[ln]-1-Pmin)+dp; t_=(T[ln]-Tmin)+dt; e_=(ee[ln]-emin)+de
          for g in groups:
            key=np.unique(np.stack([p_[g],t_[g],e_[g]],1),axis=0)
            samples.append((key,nT,nE))
print(len(samples),"group accesses; mean distinct rows %.2f"%np.mean([len(k) for k,_,_ in samples]))
def score(fn):
    tot=0
    for key,nT,nE in samples:
        p,t,e=key[:,0],key[:,1],key[:,2]
        row=(p*nT+t)*nE+e
        pos=fn(row,p,t,e,nT,nE)%16
        tot+=np.bincount(pos,minlength=16).max()
    return tot/len(samples)
print("current            %.3f"%score(lambda r,p,t,e,nT,nE:9*r))
print("xor row>>4         %.3f"%score(lambda r,p,t,e,nT,nE:9*r+((r>>4)&7)))
best=(9,None)
for a,b,c in []:
    s=score(lambda r,p,t,e,nT,nE:9*r+((a*p+b*t+c*e)%8))
    if s<best[0]: best=(s,(a,b,c)); print("rot = (%d p + %d t + %d e) mod 8: %.3f"%(a,b,c,s))
# position independent of the linear row: free choice per (p,t,e) -- what a perfect per-box colouring would reach (lower bound ~)
print("lower bound (every group conflict-free): 1.000")

# ---- minor-type accesses: rows [t][e] (no pressure dimension), stencil (t,e1),(t,e1+1),(t+1,e2),(t+1,e2+1)
msamples=[]
for t0 in range(0,ncol,512):
  for l in range(0,nlay,5):
    for b in range(16):
      gS=bl[0,b]-1; cols=slice(t0,t0+512)
      itr=(~tropo[cols,l]).astype(int)
      fl=np.where(itr==0, gf[0,gS], gf[1,gS])-1
      T=jT[cols,l]
      e1=jeta[0,cols,l,:][np.arange(512),fl]; e2=jeta[1,cols,l,:][np.arange(512),fl]
      Tmin=T.min(); emin=min(e1.min(),e2.min()); emax=max(e1.max(),e2.max())+1
      nT=T.max()+1-Tmin+1; nE=emax-emin+1
      for w in range(8):
        ln=slice(64*w,64*w+64)
        for dt,de,ee in ((0,0,e1),(0,1,e1),(1,0,e2),(1,1,e2)):
          t_=(T[ln]-Tmin)+dt; e_=(ee[ln]-emin)+de
          for g in groups:
            key=np.unique(np.stack([0*t_[g],t_[g],e_[g]],1),axis=0)
            msamples.append((key,nT,nE))
def mscore(fn, S):
    tot=0
    for key,nT,nE in S:
        p,t,e=key[:,0],key[:,1],key[:,2]
        row=(p*nT+t)*nE+e
        tot+=np.bincount(fn(row,p,t,e)%16,minlength=16).max()
    return tot/len(S)
print("minor: distinct rows per group %.2f"%np.mean([len(k) for k,_,_ in msamples]))
for name,fn in [] and (("current",lambda r,p,t,e:9*r),("xor t&1",lambda r,p,t,e:9*r+(t&1)),("xor e&1",lambda r,p,t,e:9*r+(e&1)),("xor (t&1)|(e&1)<<1 (2 bit)",lambda r,p,t,e:9*r+((t&1)|((e&1)<<1))),("xor r>>4 &1",lambda r,p,t,e:9*r+((r>>4)&1)),("xor r>>4 &7",lambda r,p,t,e:9*r+((r>>4)&7))):
    print("  minor %-28s %.3f"%(name,mscore(fn,msamples)))
for name,fn in (("current",lambda r,p,t,e:9*r),("xor p&1",lambda r,p,t,e:9*r+(p&1)),("xor (p&1)|(t&1)<<1",lambda r,p,t,e:9*r+((p&1)|((t&1)<<1))),("xor (p&1)|(e&1)<<1",lambda r,p,t,e:9*r+((p&1)|((e&1)<<1)))):
    print("  major %-28s %.3f"%(name,mscore(fn,samples)))

# ---- pure reorderings of the major box (no rotation): which dimension is innermost?  nP = 2 in nearly every box
def oscore(order):
    tot=0
    for key,nT,nE in samples:
        p,t,e=key[:,0],key[:,1],key[:,2]
        nP=int(p.max())+1 if False else 3   # boxes hold up to 3 pressure levels; the stride must be the box's own nP: use max seen
        dims={"p":(p,3),"t":(t,nT),"e":(e,nE)}
        row=0
        for d in order:
            v,n=dims[d]; row=row*n+v
        tot+=np.bincount((9*row)%16,minlength=16).max()
    return tot/len(samples)
for order in ("pte","tep","etp","tpe","ept","pet"):
    print("  major rows ordered [%s] (last = innermost): %.3f"%("][".join(order),oscore(order)))
