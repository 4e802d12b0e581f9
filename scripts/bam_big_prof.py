#!/usr/bin/env python3
"""20 dense BaM updates at (D, B) for a rocprofv3 --kernel-trace --stats table (which kernels carry a batch beyond the
one-workgroup chain).  usage: bam_big_prof.py D B"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gsmvi_amd  # noqa: E402

D, B = int(sys.argv[1]), int(sys.argv[2])
eng = gsmvi_amd.get_engine()
g = torch.Generator(device=eng.device)
g.manual_seed(1)
kw = dict(dtype=torch.float64, device=eng.device, generator=g)
mu0 = torch.randn(D, **kw)
A = torch.randn(D, D, **kw)
S0 = A @ A.T / D + 0.1 * torch.eye(D, dtype=torch.float64, device=eng.device)
S0 = (0.5 * (S0 + S0.T)).contiguous()
X = (mu0[None, :] + torch.randn(B, D, **kw) @ torch.linalg.cholesky(S0).T).contiguous()
G = (-(X - 0.3)).contiguous()
mu, S, flag = eng.empty(D), eng.empty(D, D), eng.new_flag()
for _ in range(20):
    eng.bam_update(X, G, mu0, S0, 1.0, 1e-6, out=(mu, S), flag=flag)
torch.cuda.synchronize()
print("flag", eng.read_flag(flag))
