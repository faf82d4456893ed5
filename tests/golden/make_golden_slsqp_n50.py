#!/usr/bin/env python3
"""Full-horizon cross-check of the oracle by an independent NLP solver (BASELINE configs[1] shapes: N = 50, nx = nu = 12).

    python tests/golden/make_golden_slsqp_n50.py      # ~10 minutes; writes slsqp_centroidal_n50.npz

scipy's SLSQP (a dense active-set SQP; it shares no step with the oracle's Gauss-Newton SQP + Riccati interior point) on
the NLP of one seeded centroidal problem with low friction (pyramid faces active) and a commanded forward speed: cost and
dynamics Jacobians are supplied analytically (the oracle's own A, B, which tests/test_oracle.py checks against finite
differences), 1 212 variables, 612 equality and 400 inequality constraints.  SLSQP stops after ~110 iterations with
"positive directional derivative for linesearch" -- it cannot improve the point any further at double precision; the fixture
records that point.  tests/test_oracle.py compares the oracle's converged solution with it in seconds."""
import os, sys, time; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from scipy.optimize import minimize
from oracle.oracle import Oracle
from iterative_learning_nmpc_amd import workloads as wl
o=Oracle('f64')
w=wl.centroidal_trot(B=1,N=50,seed=3)
w.mp=w.mp.copy(); w.mp[6]=0.15
w.yref=w.yref.copy(); w.yref[:,:,6]=0.8
N,nx,nu=50,12,12
b=0
W,We=np.asarray(w.W,float),np.asarray(w.W_e,float)
yref=w.yref[b]
unpack=lambda z:(np.vstack([w.x0[b],z[:N*nx].reshape(N,nx)]),z[N*nx:].reshape(N,nu))
def cost(z):
    X,U=unpack(z); e=np.hstack([X[:N],U])-yref
    return 0.5*(W*e*e).sum()+0.5*(We*(X[N]-w.yref_e[b])**2).sum()
def cost_grad(z):
    X,U=unpack(z); e=np.hstack([X[:N],U])-yref
    gx=np.zeros((N+1,nx)); gx[:N]=W[:nx]*e[:,:nx]; gx[N]=We*(X[N]-w.yref_e[b])
    gu=W[nx:]*e[:,nx:]
    return np.concatenate([gx[1:].ravel(),gu.ravel()])
def defects(z):
    X,U=unpack(z)
    return np.concatenate([o.dynamics(1,w.mp,X[k],U[k],w.params[b,k],jac=False)-X[k+1] for k in range(N)])
def defects_jac(z):
    X,U=unpack(z); J=np.zeros((N*nx,N*nx+N*nu))
    for k in range(N):
        _,A,Bm=o.dynamics(1,w.mp,X[k],U[k],w.params[b,k])
        if k>0: J[k*nx:(k+1)*nx,(k-1)*nx:k*nx]=A
        J[k*nx:(k+1)*nx,k*nx:(k+1)*nx]-=np.eye(nx)
        J[k*nx:(k+1)*nx,N*nx+k*nu:N*nx+(k+1)*nu]=Bm
    return J
Gs=[];
rows=[]
for k in range(N):
    G,h,act=o.constraints(1,w.mp,w.params[b,k])
    for j in np.nonzero(act)[0]:
        r=np.zeros(N*nx+N*nu); r[N*nx+k*nu:N*nx+(k+1)*nu]=-G[j]; rows.append((r,h[j]))
Gi=np.array([r for r,_ in rows]); hi=np.array([h for _,h in rows])
ineq=lambda z: Gi@z+hi
z0=np.concatenate([w.X[b,1:N+1].ravel(),w.U[b,:N].ravel()]).astype(float)
t=time.time()
r=minimize(cost,z0,jac=cost_grad,method='SLSQP',constraints=[dict(type='eq',fun=defects,jac=defects_jac),dict(type='ineq',fun=ineq,jac=lambda z:Gi)],options=dict(maxiter=300,ftol=1e-15))
print('slsqp',r.success,r.message,r.nit,'time',time.time()-t, 'cost',r.fun)
Xs,Us=unpack(r.x)
X,U,st,_=o.solve_batch(1,N,w.mp,o.opt(max_sqp_iter=40,n_ipm=60,tau_min=1e-10,mu0=1.0,nlp_tol=1e-10,reg=w.meta['reg'],reg_e=w.meta['reg_e'],yref_per_stage=1),w.W,w.W_e,w.x0,w.yref,w.yref_e,w.params,w.X,w.U)
print('oracle status',st,'maxdiff U',np.abs(U[0]-Us).max(),np.abs(Us).max(),'X',np.abs(X[0]-Xs).max())
G,h,act=o.constraints(1,w.mp,w.params[b,0]); print('active faces at node 0', ((G@Us[0]-h)[act>0]>-1e-6).sum())
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'slsqp_centroidal_n50.npz'), Xs=Xs, Us=Us, seed=np.int64(3), mu=np.float64(0.15), vx_ref=np.float64(0.8), slsqp_iterations=np.int64(r.nit), slsqp_cost=np.float64(r.fun))
