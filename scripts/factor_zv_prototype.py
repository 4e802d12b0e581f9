"""Numpy prototype of the round-3 factor-form algebra (basis [Z; V], scalars from the Gram matrix, K'' = S^T K S), checked against the
pinned dense oracle for random, dependent-row, exact-fixed-point and near-converged states BEFORE the kernels were written.
usage: python scripts/factor_zv_prototype.py   (CPU only; imports oracle/ as the checker)"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import gsm_oracle as orc
TOL = 64*2.220446049250313e-16
def chol_semidef(G, allow=True):
    n=G.shape[0]; A=G.copy()
    for p in range(n): A[p,p]-=TOL*A[p,p]
    R=np.zeros((n,n)); W=np.eye(n); bad=False
    S=A.copy()
    dropped=[]
    for p in range(n):
        d=S[p,p]
        if d>0 and np.isfinite(d):
            r=S[p,p:]/np.sqrt(d); R[p,p:]=r
            S[p+1:,p+1:]-=np.outer(r[1:],r[1:])
        elif allow and d<=0 and np.isfinite(d):
            dropped.append(p)
        else: bad=True
    Rm=R.copy()
    for p in dropped: Rm[p,p]=1.0
    Wg=np.linalg.inv(Rm).T
    return R,Wg,bad
def factor_update_new(Z,X,G,mu,F):
    B,D=Z.shape; n=2*B
    W=G@F.T; V=W+Z; Xc=X-mu; VF=V@F
    R1=np.vstack([Z,V]); G1=R1@R1.T
    zz=np.diag(G1)[:B]; zv=np.array([G1[b,B+b] for b in range(B)]); vv=np.diag(G1)[B:]
    zw=zv-zz; ww=vv-2*zv+zz; wv=vv-zv    # w.w, z.w, w.v=ww+zw
    rho=0.5*np.sqrt(1+4*(ww+zw**2))-0.5; den=1+rho-zw
    alpha=1/(1+rho); beta=(wv/den)/(1+rho)
    S=np.zeros((n,n)); S[:B,:B]=np.eye(B); S[B:,:B]=np.diag(beta); S[B:,B:]=np.diag(alpha)
    Gam=S@G1@S.T
    moderate=np.all(np.diag(Gam)<2**32)
    Rg,Wg,bad=chol_semidef(Gam,moderate)
    J=np.zeros((n,n)); J[:B,B:]=np.eye(B)/B; J[B:,:B]=np.eye(B)/B; J[B:,B:]=-np.eye(B)/B
    Ap=np.eye(n)+Rg@J@Rg.T
    try: T=np.linalg.cholesky(Ap).T
    except np.linalg.LinAlgError: return mu,F,False
    if bad: return mu,F,False
    Wgs=Wg@S
    K2=Wgs.T@(T-np.eye(n))@Wgs
    Tm1=np.vstack([Xc,VF])
    Fn=F+R1.T@(K2@Tm1)
    mun=mu+(beta@Xc+alpha@VF)/B
    return mun,Fn,True
def rel(a,b): return np.abs(a-b).max()/np.abs(b).max()
for (D,B,seed) in [(64,8,1),(256,32,2),(100,17,3)]:
    st=orc.make_update_state(D,B,seed); F0=st["L"].T.copy()
    mu_o,S_o=orc.gsm_update_batched(st["samples"],st["vs"],st["mu0"],st["S0"])
    mu,F,ok=factor_update_new(st["Z"],st["samples"],st["vs"],st["mu0"],F0)
    print("random",D,B,ok,rel(mu,mu_o),rel(F.T@F,S_o))
# dependent rows
for D,B in [(64,8),(256,32)]:
    rs=np.random.RandomState(D); mu0=np.zeros(D); F0=np.eye(D); Z=rs.standard_normal((B,D)); X=mu0+Z@F0
    G=-2.0*(X-0.5)
    mu_o,S_o=orc.gsm_update_batched(X,G,mu0,F0.T@F0)
    mu,F,ok=factor_update_new(Z,X,G,mu0,F0); print("dependent",D,B,ok,rel(mu,mu_o),rel(F.T@F,S_o))
    Gf=-(X-mu0); mu2,F2,ok2=factor_update_new(Z,X,Gf,mu0,F0); print(" fixed point",ok2,rel(F2.T@F2,np.eye(D)),np.abs(mu2-mu0).max())
# near converged: Gaussian target N(m,P^-1), state close
D,B=128,16
rs=np.random.RandomState(5); L=rs.standard_normal((D,D)); Sig=L@L.T/D+0.1*np.eye(D); P=np.linalg.inv(Sig); m=rs.random_sample(D)
for eps in [1e-3,1e-6,1e-9,1e-12,0.0]:
    E=rs.standard_normal((D,D)); E=(E+E.T)/2
    S0=Sig+eps*E; mu0=m+eps*rs.standard_normal(D)
    F0=np.linalg.cholesky(S0).T; Z=rs.standard_normal((B,D)); X=mu0+Z@F0; G=-(X-m)@P
    mu_o,S_o=orc.gsm_update_batched(X,G,mu0,F0.T@F0)
    mu,F,ok=factor_update_new(Z,X,G,mu0,F0)
    mf,Ff,okf=orc.gsm_factor_update(Z,G,mu0,F0.T)
    print("near conv eps",eps,ok,rel(mu,mu_o),rel(F.T@F,S_o),"| old restatement",rel(mf,mu_o),rel(Ff@Ff.T,S_o), "dist to target", rel(F.T@F,Sig))
