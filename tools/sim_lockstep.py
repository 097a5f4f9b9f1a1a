"""Step counts of the (tile, branch) kernel under other lane groupings, on REAL chain state: 64 replicas of C3 from the CPU oracle
(segment counts m before and m' after one stationary sweep, per replica and branch), priced with the measured clocks per step of
DESIGN.md section 6 (pass A step 232 + Philox, pass B step 302 + Philox, a Philox4x32-7 block 192 clocks = 48 per step when one block
serves four lock-step steps of the whole wave).

    python tools/sim_lockstep.py [n_tips]        (quarter-wave granularity -- sized before building)
Schemes:
  lockstep64     today: a wave walks its branches one after the other, pass A then pass B, each to the maximum over 64 lanes
  quarter_mixed  every 16-lane group of the wave walks the wave's branches (A then B per branch) over its own 16 replicas on its own
                 step counter, i.e. the groups drift apart by whole branches (rows stay 128-byte pieces): an iteration issues the pass-A body if any group is in pass A and the pass-B body if any is in pass B
  quarter_split  the same, but all pass-A work of a group's branches first, then all pass-B work (merged states parked in memory):
                 within a pass the groups are decoupled and only one body is issued; a group needs a Philox block every fourth of ITS
                 steps, so the block code is issued on every iteration in which some group starts a new block
  quarter_split_aligned   ... with groups starting a branch only on multiples of four iterations (one Philox block per four iterations)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from phylomap_amd import synth, treeorder  # noqa: E402

A_BODY, B_BODY, PHILOX = 232.0, 302.0, 192.0
BURN, LANES, G = 12, 64, 16


def counts(n_tips):
    z, Q, pid, Om = synth.config_problem(3, n_tips=n_tips)
    nen, nodelist, root = treeorder.pruningwiseedgeorder(z), treeorder.makenodelist(z), treeorder.myreorder(z)
    B = np.eye(4) + Q / Om
    m = np.zeros((LANES, z["edge"].shape[0]), dtype=np.int64)
    m2 = np.zeros_like(m)
    for r in range(LANES):
        for N, dst in ((BURN, m), (BURN + 1, m2)):
            _, rc, d = O.maketreelistMCMC(z, Q, pid, B, Om, nen, nodelist, root, N, variant=O.BIGTREE, seed=3, replica=r, dump=True)
            assert rc == 0
            dst[r] = d.seg_count
    lam = Om * z["edge.length"]
    return m, m2, np.argsort(-lam, kind="stable")      # branch order: longest (largest slot) first, as the kernel walks them


def lockstep64(m, m2, order):
    a, b = m.max(0), m2.max(0)
    steps = a.sum() + b.sum()
    return a.sum() * (A_BODY + PHILOX / 4) + b.sum() * (B_BODY + PHILOX / 4), steps


def quarter(m, m2, order, split, aligned=False):
    total, steps = 0.0, 0
    E = len(order)
    for w0 in range(0, E, G):                            # one wave: G consecutive branches of the sorted order; group g walks them for
        br = order[w0:w0 + G]                            # lanes 16 g .. 16 g + 15 (lanes stay replicas), on its own step counter
        groups = [br for g in range(4)]
        if split:
            for cnt, body in ((m, A_BODY), (m2, B_BODY)):
                # per group: list of per-branch step counts (max over its 16 lanes); groups run decoupled
                seqs = [[int(cnt[16 * g:16 * g + 16, b].max()) for b in groups[g]] for g in range(4)]
                if aligned:
                    seqs = [[-(-v // 4) * 4 for v in s] for s in seqs]
                lens = [sum(s) for s in seqs]
                iters = max(lens) if lens else 0
                if aligned:
                    n_blocks = iters / 4.0
                else:      # iterations in which some group starts a new block (every 4th of its own steps within a branch)
                    need = np.zeros(iters + 1, dtype=bool)
                    for s in seqs:
                        t = 0
                        for v in s:
                            need[t:t + v:4] = True
                            t += v
                    n_blocks = float(need.sum())
                total += iters * body + n_blocks * PHILOX
                steps += iters
        else:
            # A then B per branch inside a group; both bodies issued on iterations with groups in different passes
            tl = []
            for g in range(4):
                seq = []
                for b in groups[g]:
                    seq += [0] * int(m[16 * g:16 * g + 16, b].max()) + [1] * int(m2[16 * g:16 * g + 16, b].max())
                tl.append(seq)
            iters = max(len(s) for s in tl)
            for t in range(iters):
                ph = {s[t] for s in tl if t < len(s)}
                total += (A_BODY if 0 in ph else 0.0) + (B_BODY if 1 in ph else 0.0) + PHILOX      # unaligned groups: a block nearly every iteration
            steps += iters
    return total, steps


def main():
    n_tips = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    m, m2, order = counts(n_tips)
    base, bsteps = lockstep64(m, m2, order)
    print(f"C3-shaped tree, {n_tips} tips, 64 replicas after {BURN} sweeps: mean m {m.mean():.2f}, mean m' {m2.mean():.2f}, "
          f"mean over branches of max over 64 lanes {m.max(0).mean():.2f} / {m2.max(0).mean():.2f}, over 16 lanes "
          f"{np.mean([m[16 * g:16 * g + 16].max(0).mean() for g in range(4)]):.2f} / {np.mean([m2[16 * g:16 * g + 16].max(0).mean() for g in range(4)]):.2f}")
    print(f"{'lockstep64':24s} iterations {bsteps:9d}  clocks {base:14.0f}  = 1.000")
    for name, kw in (("quarter_mixed", dict(split=False)), ("quarter_split", dict(split=True)), ("quarter_split_aligned", dict(split=True, aligned=True))):
        c, st = quarter(m, m2, order, **kw)
        print(f"{name:24s} iterations {st:9d}  clocks {c:14.0f}  = {c / base:.3f}")


if __name__ == "__main__":
    main()
