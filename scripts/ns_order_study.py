#!/usr/bin/env python3
"""Would a higher-order coupled iteration need fewer LAUNCHES than the scaled Newton-Schulz step of gsmvi_bam_small.hip?
(round-3 verdict, item 1b).  Both act on the eigenvalues of M = Z Y in [l, 1]:  sigma' = sigma p(sigma^2), sigma = sqrt(lambda).
  NS, scaled:        p linear in M      2 launches per step (M = Z Y;  Y' = c Y T, Z' = c T Z)
  minimax quintic:   p quadratic in M   3 launches per step (M;  M^2;  Y p(M), p(M) Z)
This script runs the scalar recurrences (Remez for the quintic) and prints steps and launches to reach 1 - l < 1e-16.
Result: small eigenvalues grow 6.75x per NS step and 18.1x per quintic step = 2.60x against 2.63x PER LAUNCH; the final phase
is order 2 per 2 launches against order 3 per 3 (2^(1/2) = 1.41, 3^(1/3) = 1.44).  No launch is saved; not built."""
import numpy as np
np.set_printoptions(linewidth=200)
def remez_quintic(ell, iters=60):
    # odd quintic q(s)=a s+b s^3+c s^5 minimising max|1-q| on [ell,1]
    if ell >= 0.7:
        q=lambda t:(15*t-10*t**3+3*t**5)/8
        return (15/8,-10/8,3/8,-(1-q(ell)))
    # init interior points
    x = np.array([ell, ell+(1-ell)*0.3, ell+(1-ell)*0.7, 1.0])
    for _ in range(iters):
        # solve q(x_i) = 1 - (-1)^i E
        M = np.array([[xi, xi**3, xi**5, (-1)**i] for i, xi in enumerate(x)])
        a,b,c,E = np.linalg.solve(M, np.ones(4))
        # extrema: a+3b t+5c t^2=0, t=s^2
        disc = 9*b*b-20*a*c
        if disc <= 0: break
        t1 = (-3*b - np.sqrt(disc))/(10*c); t2 = (-3*b + np.sqrt(disc))/(10*c)
        ts = sorted([t for t in (t1,t2) if t>0])
        if len(ts)<2: break
        xn = np.array([ell, np.sqrt(ts[0]), np.sqrt(ts[1]), 1.0])
        if not (ell < xn[1] < xn[2] < 1.0): break
        if np.max(np.abs(xn-x)) < 1e-15: x = xn; break
        x = xn
    return a,b,c,abs(E)
def ns_scaled_steps(l):
    ks=None; out=[]
    for k in range(40):
        c2 = 3.0/(1+np.sqrt(l)+l) if l<0.25 else 1.0
        out.append(c2); x=c2*l; l=min(1.0, x*(3-x)**2/4)
        if 1-l<5e-9 and ks is None: ks=k+2
    return ks
def quintic_steps(l):
    ell = np.sqrt(l); coefs = []
    for k in range(40):
        if 1 - ell < 1e-16: break
        a, b, c, E = remez_quintic(ell)
        if E < 0:                      # Taylor quintic near convergence: [ell, 1] -> [1 + E, 1]
            coefs.append((a, b, c)); ell = 1 + E
        else:                          # minimax quintic: [ell, 1] -> [1 - E, 1 + E], rescaled to end at 1
            coefs.append((a / (1 + E), b / (1 + E), c / (1 + E))); ell = (1 - E) / (1 + E)
    return coefs
for cond in (1e2,1e4,1e6,1e8,1e10,1e12):
    l=1/cond
    cf=quintic_steps(l)
    print(f"cond {cond:g}: scaled Newton-Schulz k* = {ns_scaled_steps(l)} ({2*ns_scaled_steps(l)-1} launches)   minimax quintic {len(cf)} steps ({3*len(cf)-1} launches)"
          f"   first quintic (a, b, c) = {tuple(round(float(v), 3) for v in cf[0])}")
