import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from oracle.oracle import Oracle
from iterative_learning_nmpc_amd import workloads as wl
from iterative_learning_nmpc_amd.solver import BatchedNmpcSolver
rel=lambda a,b: float(np.linalg.norm(a-b)/np.linalg.norm(b))
B=int(sys.argv[1]) if len(sys.argv)>1 else 8
w=wl.wholebody_trot(B=B,N=30,seed=0)
s=BatchedNmpcSolver(w.model_id,w.N,B,"cuda:0")
s.set_model_params(w.mp); s.set_cost_weights(w.W,w.W_e,w.meta['reg'],w.meta['reg_e'])
o=Oracle('f64')
for n_ipm,sqp in ((0,1),(6,1),(6,3)):
    s.set_max_qp_iter(n_ipm); s.set_max_iter(sqp)
    t={k:s.to_device(getattr(w,k)) for k in ("x0","yref","yref_e","params","X","U")}
    X,U,st,stats=s.solve(t["x0"],t["yref"],t["yref_e"],t["params"],t["X"],t["U"])
    torch.cuda.synchronize()
    opt=o.opt(yref_per_stage=1,reg=w.meta['reg'],reg_e=w.meta['reg_e'],max_sqp_iter=sqp,n_ipm=n_ipm)
    Xo,Uo,sto,statso=o.solve_batch(2,w.N,w.mp,opt,w.W,w.W_e,w.x0,w.yref,w.yref_e,w.params,w.X,w.U)
    Xg,Ug=X.cpu().numpy().astype(np.float64),U.cpu().numpy().astype(np.float64)
    print('ipm',n_ipm,'sqp',sqp,'relX %.3e relU %.3e'%(rel(Xg,Xo),rel(Ug,Uo)),'status',st.cpu().numpy()[:8],sto[:8],'cost',stats[:3,0].cpu().numpy(),statso[:3,0],'step',stats[:3,1].cpu().numpy(),statso[:3,1])
    if n_ipm==0:
        # per-part errors
        print('  q %.2e v %.2e h %.2e a %.2e f %.2e'%(rel(Xg[:,:,:18],Xo[:,:,:18]),rel(Xg[:,:,18:36],Xo[:,:,18:36]),rel(Xg[:,:,36:],Xo[:,:,36:]),rel(Ug[:,:,:18],Uo[:,:,:18]),rel(Ug[:,:,18:],Uo[:,:,18:])))
