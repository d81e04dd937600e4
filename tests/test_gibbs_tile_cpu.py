"""The tile algorithm of k_gibbs_exact_coop (rsem_amd/csrc/gibbs.hip), restated in plain Python and held against
the sequential reference chain (oracle, itself pinned bit for bit on the reference's count-vector files).

The HIP kernel evaluates up to 64 consecutive reads speculatively against the counts as they were before the tile
(one read per lane) and then commits them in file order: only reads whose draw changed their transcript broadcast
(z_old, z_new); later reads holding one of the two patch their private counts and redraw with the SAME random
number.  This file checks that this schedule IS the sequential chain -- same integer count vectors -- including the
corner cases the kernel has: tiles cut short by the LDS item budget, a read larger than the whole budget (walked
alone), reads that hold the same transcript twice, the MT19937 block boundary inside a tile.  It restates the
algorithm, not the HIP code: the GPU tests compare the kernel itself with the oracle and the reference's files.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle as orc
from tools.synth_data import make_em_workload, to_gibbs_items


class Mt:
    def __init__(self, seed):
        self.L = orc.lib()
        self.buf = C.create_string_buffer(625 * 4)
        self.L.orc_mt_seed(self.buf, C.c_uint32(int(seed)))
        self.L.orc_mt_next.restype = C.c_uint32

    def next(self):
        return int(self.L.orc_mt_next(self.buf))


def tile_chain(M, rp, sid, cp, init, pseudoC, N0, seed, burnin, nsamples, gap, tile_rows=64, tile_items=2048, held_slots=1024):
    rp = rp.astype(np.int64)
    N1 = len(rp) - 1
    counts = init.astype(np.int64).copy()
    counts[0] += N0
    z = np.zeros(N1, np.int64)
    mt = Mt(seed)

    def draw(s, p, c, rnd, init_pass):
        w = p if init_pass else (c.astype(np.float64) + pseudoC) * p
        cum = np.empty(len(w))
        run = 0.0
        for k in range(len(w)):  # left to right, like one lane
            run = w[k] if k == 0 else run + w[k]
            cum[k] = run
        prb = (float(rnd) * (1.0 / 4294967296.0)) * cum[-1]
        cnt = int(np.sum(cum <= prb))
        return int(s[min(cnt, len(w) - 1)])

    def sweep(init_pass):
        i0 = 0
        while i0 < N1:
            nrt = min(tile_rows, N1 - i0)
            nr = 0
            while nr < nrt and rp[i0 + nr + 1] - rp[i0] <= tile_items:
                nr += 1
            if nr == 0:  # one read larger than the tile: walked alone
                fr, to = rp[i0], rp[i0 + 1]
                if not init_pass:
                    counts[z[i0]] -= 1
                zn = draw(sid[fr:to], cp[fr:to], counts[sid[fr:to]], mt.next(), init_pass)
                counts[zn] += 1
                z[i0] = zn
                i0 += 1
                continue
            rows = []
            for r in range(nr):  # speculative: everybody sees the counts of the tile's start
                fr, to = rp[i0 + r], rp[i0 + r + 1]
                s, p = sid[fr:to], cp[fr:to]
                c = counts[s].copy()
                zo = int(z[i0 + r])
                if not init_pass:
                    c[s == zo] -= 1
                rnd = mt.next()
                rows.append(dict(s=s, p=p, c=c, zo=zo, rnd=rnd, zn=draw(s, p, c, rnd, init_pass)))
            if init_pass:
                for r, R in enumerate(rows):
                    counts[R["zn"]] += 1
                    z[i0 + r] = R["zn"]
            else:
                # the kernel's commit filter: a changed read only takes a turn if another item of the tile carries its old
                # or new transcript (hashed counts, the read's own items included, hence >= 2)
                held = np.zeros(held_slots, np.int64)
                for R in rows:
                    np.add.at(held, R["s"] % held_slots, 1)

                def turn(R):
                    return R["zn"] != R["zo"] and (held[R["zo"] % held_slots] >= 2 or held[R["zn"] % held_slots] >= 2)

                changed = [r for r, R in enumerate(rows) if turn(R)]
                while changed:
                    r1 = changed.pop(0)
                    zo, zn = rows[r1]["zo"], rows[r1]["zn"]
                    for r in range(r1 + 1, nr):
                        R = rows[r]
                        d = (R["s"] == zn).astype(np.int64) - (R["s"] == zo).astype(np.int64)
                        if np.any(d != 0):
                            R["c"] += d
                            R["zn"] = draw(R["s"], R["p"], R["c"], R["rnd"], False)
                    changed = [r for r in range(r1 + 1, nr) if turn(rows[r])]
                for r, R in enumerate(rows):
                    if R["zn"] != R["zo"]:
                        counts[R["zo"]] -= 1
                        counts[R["zn"]] += 1
                        z[i0 + r] = R["zn"]
            i0 += nr

    sweep(True)
    out = []
    for rnd in range(1, burnin + 1 + (nsamples - 1) * gap + 1):
        sweep(False)
        if rnd > burnin and (rnd - burnin - 1) % gap == 0:
            out.append(counts.copy())
    return np.array(out, np.int32)


def _items(n_reads, seed, long_row_every=0, dup=False):
    wl = make_em_workload("tiny", seed=seed, long_row_every=long_row_every)
    rp = wl["row_ptr"][:n_reads + 1]
    nz = int(rp[-1])
    sub = dict(wl, row_ptr=rp, sid=wl["sid"][:nz].copy(), conprb=wl["conprb"][:nz], ncp=wl["ncp"][:n_reads])
    if dup:  # some reads align twice to the same transcript
        for i in range(0, n_reads, 7):
            a, b = int(rp[i]), int(rp[i + 1])
            if b - a >= 2:
                sub["sid"][a + 1] = sub["sid"][a]
    return wl["M"], to_gibbs_items(sub)


@pytest.mark.parametrize("tile_rows,tile_items,long_every,dup,held_slots", [(64, 2048, 0, False, 1024), (64, 40, 0, True, 1024), (5, 2048, 0, False, 16),
                                                                            (64, 300, 150, True, 7), (64, 2048, 0, True, 1)])
def test_tile_schedule_is_the_sequential_chain(tile_rows, tile_items, long_every, dup, held_slots):
    n = 700
    M, (irp, isid, icp) = _items(n, 11, long_row_every=long_every, dup=dup)
    init = np.zeros(M + 1, np.int32)
    N0, pseudoC = 37, 1.0
    eel, mw, grp = np.full(M + 1, 500.0), np.ones(M + 1), np.array([1, M + 1], np.int32)
    totc = (M + 1) * pseudoC + N0 + n
    burnin, nsamples, gap = 2, 3, 2
    ocv, _ = orc.gibbs_chain(M, irp, isid, icp, init, None, pseudoC, totc, N0, eel, mw, grp, 4242, burnin, nsamples, gap)
    tcv = tile_chain(M, irp, isid, icp, init, pseudoC, N0, 4242, burnin, nsamples, gap, tile_rows, tile_items, held_slots)
    assert np.array_equal(tcv, ocv)
