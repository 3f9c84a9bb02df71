import sys, numpy as np, torch
sys.path.insert(0, "cloudmicrophysics.jl_amd"); sys.path.insert(0, "tests")
import cmx
from cmx import parameters as P, synthetic
ft="f64"; dev=torch.device("cuda:0")
n_col,n_lev=2703,74
st=synthetic.sb2006_state(n_col*n_lev,dtype=torch.float64,seed=3)
cols=[c.reshape(n_col,n_lev).to(dev) for c in st]
g=torch.Generator().manual_seed(0)
inv_dz=(1.0/(30.0+470.0*torch.rand(n_lev,generator=g,dtype=torch.float64))).to(dev)
for limited in (True,False):
  for cloud in (False,True):
    mp,tps=P.Microphysics2MParams(ft,is_limited=limited),P.ThermodynamicsParameters(ft)
    got=cmx.column_tendencies_sedimentation(mp,tps,inv_dz,*cols,vel=cmx.SB2006VelType,cloud_vel=P.StokesRegimeVelType(ft) if cloud else None)
    torch.cuda.synchronize()
    for k in ("dq_lcl_dt","dn_lcl_dt","dq_rai_dt","dn_rai_dt"):
        x=getattr(got,k).reshape(-1)
        bad=torch.isnan(x).nonzero().reshape(-1)
        print(limited,cloud,k,int(bad.numel()),bad[:8].tolist())
    if cloud and not limited:
        i=1
        print([float(c.reshape(-1)[i]) for c in cols], [float(c.reshape(-1)[i+1]) for c in cols])
        v=cmx.cloud_terminal_velocity(mp.warm_rain.c.seifert_beheng.pdf_c,P.StokesRegimeVelType(ft),cols[3].reshape(-1),cols[0].reshape(-1),(cols[0]*cols[4]).reshape(-1))
        print("cloud vel nan:", int(torch.isnan(v.vt_n).sum()), int(torch.isnan(v.vt_m).sum()), v.vt_n[:4].tolist())
