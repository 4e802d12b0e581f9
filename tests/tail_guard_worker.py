"""Run by tests/test_gpu_tail_guard.py in a CHILD process: every input and output array of the engine calls sits at the very END
of its own 2 MiB device allocation, so a kernel that reads or writes past the end of an array touches the next page -- which, at
the end of a mapping, is a memory access fault that kills this process (and fails the test) instead of going unnoticed inside an
allocator block.  usage: tail_guard_worker.py D B"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import gsmvi_amd  # noqa: E402
from oracle import gsm_oracle as orc  # noqa: E402

D, B = int(sys.argv[1]), int(sys.argv[2])
SEG = int(sys.argv[3]) if len(sys.argv) > 3 else 2 * 2 ** 20      # bytes per allocation (each array ends its own allocation)
eng = gsmvi_amd.get_engine()
keep = []


def tail(shape, src=None):
    """a tensor of this shape whose last element is the last double of a fresh 2 MiB allocation"""
    n = int(np.prod(shape))
    seg = max(SEG, -(-8 * n // (2 * 2 ** 20)) * 2 * 2 ** 20)           # whole 2 MiB granules; the array ends the allocation
    big = torch.empty(seg // 8, dtype=torch.float64, device=eng.device)
    keep.append(big)
    t = big[big.numel() - n:].view(*shape)
    if src is not None:
        t.copy_(torch.as_tensor(np.ascontiguousarray(src), dtype=torch.float64))
    return t


rs = np.random.RandomState(D * 100 + B)
F0n = rs.standard_normal((D, D)) / np.sqrt(D) + 0.7 * np.eye(D)
mu0n, Zn = rs.standard_normal(D), rs.standard_normal((B, D))
Xn = mu0n + Zn @ F0n
m, _, P = orc.make_gaussian_target(D, 7)
Gn = orc.gaussian_score(Xn, m, P)
S0n = F0n.T @ F0n
for rep in range(3):                                   # (three fresh sets of allocations: three chances to abut an unmapped page)
    keep.clear()
    torch.cuda.empty_cache()
    Z, X, G, mu0, F0, S0 = tail((B, D), Zn), tail((B, D), Xn), tail((B, D), Gn), tail((D,), mu0n), tail((D, D), F0n), tail((D, D), S0n)
    Pd, md = tail((D, D), P), tail((D,), m)
    Xs = eng.sample(Z, mu0, F0, out=tail((B, D)))
    Gs = eng.gaussian_score(X, md, Pd, out=tail((B, D)))
    mu, S = eng.gsm_update(X, G, mu0, S0, out=(tail((D,)), tail((D, D))))
    R, fl = eng.potrf(S0, out=tail((D, D)))
    flags = [fl]
    if 2 * B <= D:
        _, _, f1 = eng.gsm_factor_update(Z, X, G, mu0, F0, out=(tail((D,)), tail((D, D))))
        _, _, f2 = eng.bam_factor_update(Z, X, G, mu0, F0, 1.5, out=(tail((D,)), tail((D, D))))
        flags += [f1, f2]
    _, _, f3 = eng.bam_update(X, G, mu0, S0, 1.5, 1e-6, out=(tail((D,)), tail((D, D))))
    flags.append(f3)
    torch.cuda.synchronize()
    assert all(eng.read_flag(f) == 0 for f in flags)
    mu_o, S_o = orc.gsm_update_batched(Xn, Gn, mu0n, S0n)
    assert np.abs(S.cpu().numpy() - S_o).max() <= 1e-8 * np.abs(S_o).max()       # (parity proper is elsewhere: this test is about faults)
    assert np.abs(Xs.cpu().numpy() - Xn).max() <= 1e-9 * np.abs(Xn).max()
    assert np.abs(Gs.cpu().numpy() - Gn).max() <= 1e-8 * np.abs(Gn).max()
print("tail guard ok", D, B)
