import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, gsmvi_amd
from oracle import gsm_oracle as orc, bam_oracle as borc
from conftest import rel_err
eng = gsmvi_amd.get_engine()
D = 160
for B, scale, reg in ((4, 1e2, 1e3), (4, 1e3, 1e3), (4, 1e4, 1e3), (60, 3e3, 1e3), (128, 1e3, 1e3), (16, 1e5, 1e3), (32, 1.0, 1e7), (32, 30.0, 100.0)):
    st = orc.make_update_state(D, B, seed=B)
    Gs = st["vs"] * scale
    X, G, mu0, S0 = (eng.asarray(a) for a in (st["samples"], Gs, st["mu0"], st["S0"]))
    mu_d, S_d, f_d = eng.bam_update(X, G, mu0, S0, reg, 0.0)
    eng.set_tuning("bam_host", 1)
    mu_h, S_h, f_h = eng.bam_update(X, G, mu0, S0, reg, 0.0)
    eng.set_tuning("bam_host", 0)
    mu_o, S_o = borc.bam_lowrank_update_exact(st["samples"], Gs, st["mu0"], st["S0"], reg)
    S_o = 0.5 * (S_o + S_o.T)
    print(f"B={B} scale={scale:g} reg={reg:g}: flags {eng.read_flag(f_d)} {eng.read_flag(f_h)} | dev-vs-oracle mu {rel_err(mu_d.cpu().numpy(), mu_o):.1e} S {rel_err(S_d.cpu().numpy(), S_o):.1e} | host-vs-oracle mu {rel_err(mu_h.cpu().numpy(), mu_o):.1e} S {rel_err(S_h.cpu().numpy(), S_o):.1e}")
