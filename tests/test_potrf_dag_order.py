"""CPU: the task order of k_potrf_dag (gsm-vi_amd/csrc/gsmvi_potrf.hip) cannot deadlock.

The kernel's workers draw tickets in order and each BLOCKS (bounded) on what its tile needs; the chain workgroup blocks on the
tiles the workers leave for it.  The comment in the kernel argues: a task waits only for tiles with smaller tickets or for the
chain, and chain iteration c waits only for tiles of rows <= c that precede every tile waiting for it.  This test restates the
ticket order and the wait sets from the kernel's own lines (checked to be present in the source) and
  * checks the claim literally for every task of every grid size up to 40 block rows (and a few up to 192), and
  * SIMULATES the launch with 1, 2, 3 and 7 blocking workers: every workgroup finishes.
A change of the order in the kernel that is not made here fails the source check; a change made in both that can deadlock fails
the simulation.  (It did: with W_c published by the NEXT chain iteration and the diagonal tile at the head of its own row, the
solves of row c -- tickets below that tile's -- waited for an iteration that waited for the tile: a grid with fewer workers than
a row has solves, D >= 16448 on 256 CUs, would have stopped until the poll budget ran out.  The tile now precedes them.)"""
import os

import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "gsm-vi_amd", "csrc", "gsmvi_potrf.hip")

# the lines this file restates
KERNEL_LINES = [
    "hasC = (row >= 1 && row + 1 < nblk) ? 1 : 0;          // the two tiles chain iteration row + 1 waits for",
    "const int nG = nblk - row - 2 > 0 ? nblk - row - 2 : 0;",
    "const int cnt = 2 * hasC + nG;",
    "const int kind = (r_ < hasC) ? 1 : (r_ < 2 * hasC ? 0 : 2);",
    "const int I = (kind == 0) ? row + 1 : row;",
    "const int J = (kind == 2) ? row + 2 + (r_ - 2 * hasC) : row + 1;",
    "const int P = (kind == 0) ? I - 1 : I;                    // rank-64 updates this task applies",
    "if (!dag_wait(flags, wready + I, 1, nullptr, 0, nullptr, 0, &sh_w, max_spin)) return;",
    "dag_publish(tstep + I * nblk + J, kind == 0 ? I - 1 : I);",
    "dag_publish(xready + J * nblk + I, 1);",
    "if (!dag_wait(flags, tstep + (cI - 1) * nblk + cI, cI - 1, tstep + cI * nblk + cI, cI - 1, nullptr, 0, &sh_w, max_spin)) {",
    "if (cI > 0) dag_publish(xready + cI * nblk + (cI - 1), 1);",
    "if (tid == 0) dag_st(wready + (cI - 1), 1);",
    "ntasks += 2 * (r >= 1 && r + 1 < nblk) + (nblk - r - 2 > 0 ? nblk - r - 2 : 0);",
]


def test_the_restated_lines_are_the_kernels():
    src = open(SRC).read()
    for ln in KERNEL_LINES:
        assert ln in src, f"k_potrf_dag changed ({ln!r}): restate tests/test_potrf_dag_order.py"


def tasks(nblk):
    """Ticket order: (I, J, kind) -- kind 0: diagonal tile for the chain, 1: tile (I, I+1) for the chain, 2: tile solved by the worker."""
    out = []
    for row in range(nblk):
        if row >= 1 and row + 1 < nblk:
            out.append((row, row + 1, 1))
            out.append((row + 1, row + 1, 0))
        for J in range(row + 2, nblk):
            out.append((row, J, 2))
    return out


def needs(task):
    """What a worker task blocks on: ('x', p, J) = solved block (p, J) in R, ('w', I) = W_I published."""
    I, J, kind = task
    P = I - 1 if kind == 0 else I
    n = set()
    for p in range(P):
        n.add(("x", p, I))
        n.add(("x", p, J))
    if kind == 2:
        n.add(("w", I))
    return n


def gives(task):
    I, J, kind = task
    return {("t", I, J)} if kind != 2 else {("x", I, J)}


def chain_needs(c):
    return {("t", c - 1, c), ("t", c, c)} if c >= 2 else set()


def chain_gives(c, nblk):
    # iteration c publishes W_{c-1} (behind its loads) and the solved block (c-1, c); W_{nblk-1} has no reader
    g = set()
    if c >= 1:
        g |= {("w", c - 1), ("x", c - 1, c)}
    return g


@pytest.mark.parametrize("nblk", list(range(1, 41)) + [64, 96, 128, 192])
def test_every_wait_is_for_a_smaller_ticket_or_for_the_chain(nblk):
    ts = tasks(nblk)
    assert len(ts) == sum(2 * (r >= 1 and r + 1 < nblk) + max(nblk - r - 2, 0) for r in range(nblk))
    producer = {}
    for k, t in enumerate(ts):
        for g in gives(t):
            assert g not in producer
            producer[g] = ("task", k)
    for c in range(nblk):
        for g in chain_gives(c, nblk):
            assert g not in producer
            producer[g] = ("chain", c)
    for k, t in enumerate(ts):
        for n in needs(t):
            kind, idx = producer[n]                               # (KeyError = a wait nobody satisfies)
            if kind == "task":
                assert idx < k, (nblk, t, n, ts[idx])
            else:
                # chain iteration idx must itself need only tickets below k
                for c in range(idx + 1):
                    for cn in chain_needs(c):
                        pk, pidx = producer[cn]
                        assert pk == "task" and pidx < k, (nblk, t, n, c, cn)
    for c in range(nblk):
        for cn in chain_needs(c):
            assert producer[cn][0] == "task"


@pytest.mark.parametrize("nblk,workers", [(n, w) for n in (2, 3, 4, 5, 8, 16, 33) for w in (1, 2, 3, 7)])
def test_blocking_workers_all_finish(nblk, workers):
    ts = tasks(nblk)
    done = set()
    ticket = 0
    cur = [None] * workers                                        # the task each worker holds (blocked or about to run)
    chain_c = 0
    finished = 0
    for _round in range(10 * (len(ts) + nblk) + 10):
        progress = False
        for w in range(workers):
            if cur[w] is None and ticket < len(ts):
                cur[w] = ts[ticket]
                ticket += 1
                progress = True
            if cur[w] is not None and needs(cur[w]) <= done:
                done |= gives(cur[w])
                cur[w] = None
                finished += 1
                progress = True
        if chain_c < nblk and chain_needs(chain_c) <= done:
            done |= chain_gives(chain_c, nblk)
            chain_c += 1
            progress = True
        if finished == len(ts) and chain_c == nblk:
            return
        assert progress, f"deadlock: nblk {nblk}, {workers} workers, chain at {chain_c}, holding {cur}"
    raise AssertionError("did not finish")
