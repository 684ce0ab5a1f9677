import sys; sys.path.insert(0,".")
import numpy as np, torch
from rte_rrtmgp_amd import hiplib, frontend, synth
sys.path.insert(0,"tests")
if len(sys.argv)>1: hiplib.LIB_NAMES["dp"]=sys.argv[1]
hip=hiplib.load()
from oracle import oracle as orc
oc=orc.load_c()
kd=synth.make_kdist("lw"); ncol,nlay=700,20
atm=synth.make_atmosphere(ncol,nlay,seed=5,kdist=kd)
xp=frontend.TorchArrays("cuda:0"); go=frontend.GasOptics(hip,kd,xp); A=xp.asarray
play,tlay,col_gas=A(atm.play),A(atm.tlay),A(atm.col_gas)
st=go.interpolation(ncol,nlay,play,tlay,col_gas)
npx=frontend.NumpyArrays(); go2=frontend.GasOptics(oc,kd,npx)
st2=go2.interpolation(ncol,nlay,atm.play,atm.tlay,atm.col_gas)
ref=np.zeros((ncol,nlay,kd.ngpt),order="F") if False else npx.full((ncol,nlay,kd.ngpt),0.0)
go2.compute_tau_absorption(ncol,nlay,st2,atm.play,atm.tlay,atm.col_gas,ref)
for defer in (1,0):
    hiplib.ext_call(hip,"rte_hip_defer_zero",["i"],defer)
    tau=xp.full((ncol,nlay,kd.ngpt),3.0); hip.zero_array_3D(ncol,nlay,kd.ngpt,tau)
    go.compute_tau_absorption(ncol,nlay,st,play,tlay,col_gas,tau)
    t=xp.to_numpy(tau) if hasattr(xp,"to_numpy") else tau.cpu().numpy()
    t=np.asarray(t).reshape(ref.shape, order="F") if t.shape!=ref.shape else t
    err=np.abs(t-ref)/np.abs(ref)
    bad=np.argwhere(err>1e-12)
    print("defer",defer,"max rel",err.max(),"nbad",len(bad), bad[:5].tolist(), "cols", sorted(set(bad[:,0].tolist()))[:10], "gs", sorted(set(bad[:,2].tolist()))[:20], "lays", sorted(set(bad[:,1].tolist()))[:20])
