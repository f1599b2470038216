import sys, time, json
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, scipy.linalg as sla
from oracle import gdml_oracle as orc
from _pcg_compare import crossings
g=dict(np.load('/root/repo/tests/golden/pcg_n12_p6_m200.npz',allow_pickle=True))
M,N=g['R_train'].shape[:2]
xd,gd=orc.desc_from_R(g['R_train'].reshape(M,-1))
tp=orc.tril_perms_from_atom_perms(g['perms']); lin=orc.tril_perms_lin_from_tril_perms(tp)
sig,lam,y,idx=float(g['sig']),float(g['lam']),g['y'],g['inducing_pts_idxs']
t=time.time(); K=orc.assemble_K(xd,gd,lin,sig); print('K',time.time()-t,flush=True)
np.save('/tmp/bisect/K.npy',K)
fac=orc.nystroem_factor(xd,gd,lin,sig,lam,idx); np.save('/tmp/bisect/fac.npy',fac)
ny=np.linalg.norm(y); lv=(0.3,0.1,0.03,0.01,3e-3,1e-3)
def run(A,P,name):
    h=[]
    x,info,it,res=orc.pcg(A,y,M_mv=lambda r:(h.append(np.linalg.norm(r)),P(r))[1],rtol=1e-4,maxiter=5000)
    hist=np.array(h[1:]+[res]); print(name,it,crossings(hist,ny,lv).tolist(),flush=True); return hist
out={}
out['ref']=g['resid_hist']; print('reference',len(out['ref']),crossings(out['ref'],ny,lv).tolist())
P=lambda r: orc.precon_apply(fac,lam,r)
out['dense']=run(lambda v:-(K@v-lam*v),P,'dense K@v + oracle precon')
out['contract']=run(lambda v:-orc.kernel_matvec(xd,gd,tp,sig,lam,v),P,'oracle contraction matvec + oracle precon')
np.savez('/tmp/bisect/cpu_hist.npz',**out)
