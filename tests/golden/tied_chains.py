#!/usr/bin/env python3
"""Which golden cases hold a read whose printed chain depends on the ORDER of its alignments - the reference keeps them in a std::set ordered by heap address
(chaining.cpp:201), so a program that links the reference's chaining (oracle/_ref/mTR_ref_gpu: the binding of INTEGRATION.md) may print another
chain of the same score for such a read.  A case is listed when, for some read, the sweep of chaining.cpp:243-363 (restated below like mtr_amd/host/chain.c) picks
different records for different orders of the read's inserted records (golden G4: identity, reversed, eight seeded shuffles).  Every case NOT listed must come out
of the reference's front end byte for byte (tests/test_gpu_cli.py).

  python tests/golden/tied_chains.py      -> tests/golden/tied_chain_cases.json
"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OVERLAP = 10


def chain(recs):
    """recs: [(start, end, matches)] in insertion order -> indices of the printed chain (mtr_amd/host/chain.c: mtrh_chain)"""
    n = len(recs)
    if n == 0:
        return []
    if n == 1:
        return [0] if recs[0][0] + OVERLAP <= recs[0][1] else []
    score = [r[2] for r in recs]
    pred = [-1] * n
    ev = []
    for i, (s, e, _) in enumerate(recs):
        if s + OVERLAP <= e:
            ev.append((s, len(ev), i, True))
            ev.append((e - OVERLAP, len(ev), i, False))
    ev.sort(key=lambda t: (t[0], t[1]))
    Y = []
    for key, _, a, _ in ev:
        s, e, _ = recs[a]
        if key == s:                                        # Alignment::isStart (a start whose key equals its end key counts as a start, as in chain.c)
            p = -1
            for y in Y:
                if recs[y][1] <= s + OVERLAP:
                    p = y
                else:
                    break
            if p >= 0:
                pred[a] = p
                score[a] += score[p]
        else:
            if any(recs[y][1] <= e and score[y] > score[a] for y in Y):
                continue
            pos = len(Y)
            while pos > 0 and recs[Y[pos - 1]][1] > e:
                pos -= 1
            Y.insert(pos, a)
            t = 0
            while t < len(Y):
                if recs[Y[t]][1] >= e and score[Y[t]] < score[a]:
                    del Y[t]                                  # (the reference's erase loop skips the element behind an erased one)
                t += 1
    out = []
    if Y:
        a = Y[-1]
        while a >= 0:
            out.append(a)
            a = pred[a]
    return out[::-1]


def order_sensitive(recs, lines):
    """lines[i] = what record i prints: two orders that pick different but identical-looking records print the same report"""
    base = sorted(lines[i] for i in chain(recs))
    rng = random.Random(len(recs) * 7919 + sum(r[2] for r in recs))
    perms = [list(range(len(recs)))[::-1]] + [rng.sample(range(len(recs)), len(recs)) for _ in range(8)]
    for perm in perms:
        got = sorted(lines[perm[i]] for i in chain([recs[j] for j in perm]))
        if got != base:
            return True
    return False


def main():
    from tests import golden_util as gu
    names = sorted({f.split(".")[0] for f in os.listdir(gu.GOLDEN) if f.endswith(".default.cap.jsonl.gz")})
    out = {}
    for name in names:
        for mode in ("default", "p"):
            try:
                cap = gu.capture_by_read(name, mode)
            except Exception:
                continue
            tied = 0
            for per_read in cap:
                g4 = [gu.g4_tuple(ev) for ev in per_read["G4"]]
                recs = [(t[0], t[1], t[5]) for t in g4]
                lines = [(t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8], t[13]) for t in g4]
                if len(recs) > 1 and order_sensitive(recs, lines):
                    tied += 1
            out[f"{name}.{mode}"] = tied
    res = {"made_by": "tests/golden/tied_chains.py", "reads_with_an_order_dependent_chain": {k: v for k, v in out.items() if v}, "cases_checked": sorted(out)}
    with open(os.path.join(gu.GOLDEN, "tied_chain_cases.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res["reads_with_an_order_dependent_chain"]))


if __name__ == "__main__":
    main()
