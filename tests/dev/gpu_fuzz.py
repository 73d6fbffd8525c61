#!/usr/bin/env python3
"""Development aid (GPU box): pathological reads against the CPU oracle — homopolymers, units at the period limit (450-520
bases), nested and adjacent repeats, skewed base composition, every length 1..40, repeats flush with either end, one short
unit over a whole long read.  Both DI modes, both kernel modes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from concurrent.futures import ProcessPoolExecutor
import numpy as np
from mtr_amd import synth


def oracle_chunk(args):
    manhattan, reads = args
    from tests.oracle_binding import Oracle
    o = Oracle(manhattan=manhattan)
    out = [o.process(r) for r in reads]
    o.close()
    return out


def make(seed, scale):
    rng = np.random.RandomState(seed)
    R = lambda n: rng.randint(0, 4, size=n).astype(np.uint8)
    reads = []
    for L in range(1, 41):
        reads.append(R(L)); reads.append(np.full(L, L % 4, np.uint8))
    for b in range(4):
        for L in (64, 100, 999, 1000, 1001, 3000):
            reads.append(np.full(L, b, np.uint8))
    for _ in range(40 * scale):                                   # units around MAX_PERIOD
        u = int(rng.randint(440, 530)); c = int(rng.randint(6, 12))
        body, _ = synth.make_read(rng, u, c, int(rng.randint(0, 300)), int(rng.randint(0, 300)), profile=[(0, 0, 0), (0.5, 1, 1), synth.NANOPORE][rng.randint(0, 3)])
        reads.append(body)
    for _ in range(60 * scale):                                   # nested: a unit made of a sub-repeat plus a spacer
        sub = R(int(rng.randint(2, 6))); unit = np.concatenate([np.tile(sub, int(rng.randint(3, 8))), R(int(rng.randint(1, 12)))])
        body = np.tile(unit, int(rng.randint(6, 30)))
        noise = rng.rand(len(body)) < 0.03
        body = np.where(noise, R(len(body)), body).astype(np.uint8)
        reads.append(np.concatenate([R(int(rng.randint(0, 200))), body, R(int(rng.randint(0, 200)))]))
    for _ in range(60 * scale):                                   # two or three adjacent repeats, no spacer
        parts = []
        for _k in range(int(rng.randint(2, 4))):
            parts.append(synth.make_read(rng, int(rng.choice([2, 3, 4, 5, 7, 11, 24, 60, 130])), int(rng.randint(6, 40)), 0, 0, profile=(1, 2, 2))[0])
        reads.append(np.concatenate(parts))
    for _ in range(60 * scale):                                   # skewed composition
        p = rng.dirichlet([0.3, 0.3, 0.3, 0.3]); L = int(rng.randint(50, 4000))
        reads.append(rng.choice(4, size=L, p=p).astype(np.uint8))
    for _ in range(20 * scale):                                   # one short unit over a whole long read
        u = R(int(rng.randint(1, 9))); L = int(rng.randint(3000, 20000))
        body = np.tile(u, L // len(u) + 1)[:L]
        noise = rng.rand(L) < rng.choice([0, 0.01, 0.05])
        reads.append(np.where(noise, R(L), body).astype(np.uint8))
    for _ in range(60 * scale):                                   # flush with an end, cut inside a copy
        u = int(rng.choice([2, 3, 5, 8, 13, 21, 34, 55, 89, 144])); body, _ = synth.make_read(rng, u, int(rng.randint(6, 30)), 0, 0, profile=(1, 3, 2))
        body = body[int(rng.randint(0, u)): len(body) - int(rng.randint(0, u))]
        reads.append(np.concatenate([body, R(int(rng.randint(0, 500)))]) if rng.randint(0, 2) else np.concatenate([R(int(rng.randint(0, 500))), body]))
    return reads


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    import mtr_amd
    bad_total = 0
    with ProcessPoolExecutor(max_workers=14) as pool:
        for seed, manhattan in ((1, True), (2, False), (3, True)):
            reads = make(seed, scale)
            order = np.argsort([-len(r) for r in reads])
            chunks = [[reads[i] for i in order[j::56]] for j in range(56)]
            t0 = time.time()
            res = list(pool.map(oracle_chunk, [(manhattan, c) for c in chunks]))
            want = [None] * len(reads)
            for j, ch in enumerate(res):
                for i, w in zip(order[j::56], ch):
                    want[i] = w
            t_or = time.time() - t0
            # "batches of 100": the range finder of few reads (one wavefront per (read, pass, segment), the passes of a read as a
            # pipeline of 16 wavefronts, mtr_k1_finish_pipe) instead of one wavefront per read
            modes = {"per-read": dict(MTR_STAGED="0", MTR_QUAD_MIN="0", MTR_TWO_PASS="0"), "staged": dict(MTR_STAGED="1", MTR_QUAD_MIN="0", MTR_TWO_PASS="0"),
                     "staged+quads": dict(MTR_STAGED="1", MTR_QUAD_MIN="1", MTR_TWO_PASS="0"), "staged, batches of 100": dict(MTR_STAGED="1", MTR_QUAD_MIN="0", MTR_TWO_PASS="0"),
                     "staged+quads, two passes": dict(MTR_STAGED="1", MTR_QUAD_MIN="1", MTR_TWO_PASS="1"),
                     "staged, two passes, batches of 100": dict(MTR_STAGED="1", MTR_QUAD_MIN="0", MTR_TWO_PASS="1")}
            for split in modes:
                os.environ.update(modes[split])
                eng = mtr_amd.Engine(manhattan=manhattan)
                t0 = time.time()
                if "batches" in split:
                    got = []
                    for b in range(0, len(reads), 100):
                        got += eng.process(reads[b:b + 100])
                else:
                    got = eng.process(reads)
                t_gpu = time.time() - t0
                bad = [i for i in range(len(reads)) if [tuple(r) for r in got[i]] != want[i]]
                bad_total += len(bad)
                print(f"seed {seed} {'manhattan' if manhattan else 'pearson  '} {split}: {len(reads)} reads ({sum(map(len, reads)) / 1e6:.1f} Mb, "
                      f"{sum(len(w) for w in want)} records), {len(bad)} differ (gpu {t_gpu:.2f} s, oracle pool {t_or:.1f} s)"
                      f"{' first: ' + str([(i, len(reads[i])) for i in bad[:5]]) if bad else ''}", flush=True)
                eng.close()
    print("TOTAL MISMATCHES:", bad_total)
    sys.exit(1 if bad_total else 0)


if __name__ == "__main__":
    main()
