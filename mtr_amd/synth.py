"""Seeded synthetic tandem-repeat reads (the benchmark input spec of SURVEY.md §8d).

Restates the *distribution* of the reference's generator test_single_TR/util/rand_seq.cpp:48-222
(which seeds from random_device, so its files are not reproducible): per read draw a unit of
length u uniformly over ACGT, reject it if it is a power of a shorter string (:126-170), repeat it
c times, choose exactly round(u*c*p) distinct positions for each of substitution (forced different
base), insertion (a random base *after* the kept base) and deletion (:55-57, :82-122, :176-213),
and put uniformly random flanks before and after.  ID = decimal index, one sequence line per read.

Nanopore error profile of test_single_TR/test.sh:12-14: sub 1.6 %, ins 9.0 %, del 3.8 %.
"""
from __future__ import annotations

import numpy as np

NANOPORE = (1.6, 9.0, 3.8)
# the two alternates the reference's script carries as comments (test_single_TR/test.sh:12-18): substitution-heavy profiles
PROFILE_SUB_HEAVY = (12.7, 3.2, 4.7)
PROFILE_SUB_DEL = (9.7, 2.9, 7.5)
PROFILES = {"nanopore": NANOPORE, "sub_heavy": PROFILE_SUB_HEAVY, "sub_del": PROFILE_SUB_DEL}
_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def _is_power_of_shorter(unit: np.ndarray) -> bool:
    u = len(unit)
    for d in range(1, u):
        if u % d == 0 and np.array_equal(np.tile(unit[:d], u // d), unit):
            return True
    return False


def make_read(rng: np.random.RandomState, unit_len: int, copies: int, pre: int, post: int,
              profile=NANOPORE):
    """Returns (codes uint8[L] with 0..3 = A C G T, unit codes uint8[unit_len])."""
    while True:
        unit = rng.randint(0, 4, size=unit_len).astype(np.uint8)
        if not _is_power_of_shorter(unit):
            break
    rep_len = unit_len * copies
    n_sub, n_ins, n_del = (int(round(rep_len * p / 100.0)) for p in profile)
    perm = rng.permutation(rep_len)
    kind = np.zeros(rep_len, dtype=np.uint8)
    kind[perm[:n_sub]] = 1
    kind[perm[n_sub:n_sub + n_ins]] = 2
    kind[perm[n_sub + n_ins:n_sub + n_ins + n_del]] = 3
    body = np.tile(unit, copies)
    sub = (body + rng.randint(1, 4, size=rep_len).astype(np.uint8)) % 4      # forced different base
    first = np.where(kind == 1, sub, body)
    extra = rng.randint(0, 4, size=rep_len).astype(np.uint8)
    # emit: deletion -> nothing, insertion -> base + extra, else one base
    counts = np.where(kind == 3, 0, np.where(kind == 2, 2, 1))
    out = np.empty(int(counts.sum()), dtype=np.uint8)
    pos = np.cumsum(counts) - counts
    keep = kind != 3
    out[pos[keep]] = first[keep]
    insm = kind == 2
    out[pos[insm] + 1] = extra[insm]
    left = rng.randint(0, 4, size=pre).astype(np.uint8)
    right = rng.randint(0, 4, size=post).astype(np.uint8)
    return np.concatenate([left, out, right]), unit


CONFIGS = {
    # name: (unit_len or (lo,hi), copies, flank or None(=fit to ~2 kb), n_reads, seed)
    "c2": (100, 10, 100, 1000, 1),          # BASELINE.json configs[1]: L~1.25 kb
    "headline2k": (100, 10, 500, 10000, 2),  # the "2 kb" headline: L~2.05 kb
    "c3": (200, 200, 200, 100, 3),          # configs[2] shape: L~42 kb
    "c4": ((50, 200), 10, None, 100000, 4),  # configs[3]: mixed unit lengths, L~2 kb
}


def make_reads(config: str, n_reads: int | None = None, seed: int | None = None, profile=NANOPORE):
    """List of (id_str, codes) for a named config; n_reads/seed override the defaults; profile = (sub, ins, del) in percent."""
    unit, copies, flank, n_def, seed_def = CONFIGS[config]
    n = n_def if n_reads is None else n_reads
    rng = np.random.RandomState(seed_def if seed is None else seed)
    reads = []
    for i in range(n):
        reads.append((str(i), _one_read(rng, unit, copies, flank, profile)))
    return reads


def _one_read(rng, unit, copies, flank, profile=NANOPORE):
    u = unit if isinstance(unit, int) else int(rng.randint(unit[0], unit[1] + 1))
    f = flank if flank is not None else max(0, (2052 - u * copies) // 2)
    return make_read(rng, u, copies, f, f, profile)[0]


def make_rng_checkpoints(config: str, n_reads: int, every: int, seed: int | None = None):
    """(indices, MT19937 keys, positions) of make_reads(config)'s generator before reads 0, every, 2 every, ...: what lets a rank
    of a sharded job generate ITS block of the stream without generating the reads before it (make_reads_range)."""
    unit, copies, flank, _, seed_def = CONFIGS[config]
    rng = np.random.RandomState(seed_def if seed is None else seed)
    idx, keys, pos = [], [], []
    for i in range(n_reads):
        if i % every == 0:
            st = rng.get_state()
            idx.append(i); keys.append(np.array(st[1], dtype=np.uint32)); pos.append(int(st[2]))
        _one_read(rng, unit, copies, flank)
    return np.array(idx, np.int64), np.stack(keys), np.array(pos, np.int64)


def make_reads_range(config: str, lo: int, hi: int, seed: int | None = None, checkpoints=None):
    """Reads lo .. hi-1 of make_reads(config, n >= hi, seed), as (id_str, codes).  With `checkpoints` (make_rng_checkpoints of the
    same config and seed) the generator starts at the last checkpoint at or before lo instead of at read 0."""
    unit, copies, flank, _, seed_def = CONFIGS[config]
    rng = np.random.RandomState(seed_def if seed is None else seed)
    start = 0
    if checkpoints is not None:
        idx, keys, pos = checkpoints
        k = int(np.searchsorted(idx, lo, side="right")) - 1
        if k >= 0:
            rng.set_state(("MT19937", keys[k], int(pos[k]), 0, 0.0))
            start = int(idx[k])
    out = []
    for i in range(start, hi):
        codes = _one_read(rng, unit, copies, flank)
        if i >= lo:
            out.append((str(i), codes))
    return out


def make_mixed_file(n_reads: int, seed: int, max_len: int = 12000):
    """Reads of widely different lengths (200 b .. max_len) in one file, the repeat anywhere including flush with
    either end: the input on which the reference's results depend on the ORDER of the reads —
    used by the tests of the file-order mode."""
    rng = np.random.RandomState(seed)
    reads = []
    for i in range(n_reads):
        u = int(rng.choice([2, 3, 5, 7, 12, 20, 33, 50, 100, 150]))
        copies = int(rng.randint(4, 60))
        total = int(np.exp(rng.uniform(np.log(200), np.log(max_len))))
        room = max(0, total - u * copies)
        pre = int(rng.randint(0, room + 1)) if rng.randint(0, 4) else (0 if rng.randint(0, 2) else room)
        codes, _ = make_read(rng, u, copies, pre, room - pre)
        reads.append((f"m{i}", codes))
    return reads


def write_fasta(path: str, reads) -> None:
    with open(path, "wb") as fh:
        for rid, codes in reads:
            fh.write(b">" + rid.encode() + b"\n")
            fh.write(_BASES[codes].tobytes() + b"\n")


if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("config", choices=sorted(CONFIGS))
    ap.add_argument("out")
    ap.add_argument("-n", type=int, default=None)
    ap.add_argument("--seed", type=int, default=None)
    a = ap.parse_args()
    write_fasta(a.out, make_reads(a.config, a.n, a.seed))
