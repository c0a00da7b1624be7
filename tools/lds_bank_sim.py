#!/usr/bin/env python3
"""LDS bank-conflict model of the K2t table accesses (no GPU): cycles per wave instruction from the lane groups and
bank mapping of /opt/skills/guides/MI355X_MICROARCH.md (LDS section), for random slots (a hashed table) and for
ascending slots with the gaps a half-loaded ordered table has.  HISTORY.md, K2t: why the ordered table was not built."""
import numpy as np
rng = np.random.default_rng(1)
G128 = [ [0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27], [4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31],
         [32,33,34,35,44,45,46,47,52,53,54,55,56,57,58,59], [36,37,38,39,40,41,42,43,48,49,50,51,60,61,62,63] ]
G32 = [list(range(32)), list(range(32,64))]
def cycles(addrs, width, groups, nbanks, active=None):
    # addrs: byte addresses per lane (64); width bytes; returns LDS cycles (sum over groups of max distinct addresses per bank)
    tot = 0
    for g in groups:
        banks = {}
        for l in g:
            if active is not None and not active[l]: continue
            a = addrs[l]
            for w in range(width // 4):
                b = ((a // 4) + w) % nbanks
                banks.setdefault(b, set()).add((a // 4 + w))
        tot += max((len(s) for s in banks.values()), default=0) if banks else 0
        if not banks: tot += 0
    return tot
slots = 2048
def trial(kind, n=2000):
    acc = {}
    for _ in range(n):
        if kind == 'random':
            s0 = rng.integers(0, slots // 2, 64) * 2; s1 = rng.integers(0, slots // 2, 64) * 2
        else:
            # ordered: 64 ascending slots, gaps geometric mean 3.2 slots (98 rows over 205 slots window at 50% load, 65% presence)
            gaps = rng.geometric(1 / 3.2, 64); s0 = (rng.integers(0, slots) + np.cumsum(gaps)) % slots; s0 = s0 // 2 * 2
            s1 = rng.integers(0, slots // 2, 64) * 2
        r = {}
        r['b128 bucket0'] = cycles(s0 * 8, 16, G128, 64)
        r['b128 bucket1'] = cycles(s1 * 8, 16, G128, 64)
        slot = s0 + rng.integers(0, 2, 64)
        r['add u32 interleaved (same half)'] = cycles(slot * 8, 4, G32, 32)
        r['add u32 split arrays'] = cycles(slot * 4, 4, G32, 32)
        r['b64 key read (32-lane groups, 64 banks)'] = cycles(slot * 8, 8, G32, 64)
        r['16B slot {key,sums} b128'] = cycles(slot * 16, 16, G128, 64)
        for k, v in r.items(): acc[k] = acc.get(k, 0) + v
    return {k: v / n for k, v in acc.items()}
for kind in ('random', 'ordered'):
    print(kind, {k: round(v, 2) for k, v in trial(kind).items()})

# ---- round 6: the order-preserving slot function priced per ROUND of 64 records, as the kernel would
# run it -- bucket 0 = floor((key - tile_lo) x scale), bucket 1 still hashed (the two-choice table keeps its capacity), or
# ONE ordered bucket of four slots (two adjacent ds_read_b128) -- for the run lengths of the shapes that matter, and with
# the lanes dealt to records so that each ds_read_b128 lane group holds 16 CONSECUTIVE records (free: a per-lane constant
# offset of the record loads).  LDS cycles per round = 2 bucket reads + the count's add; today's kernel: random / random.
def round_cycles(run_len, rows_per_tile=1024, nb=2048, presence=0.65, scheme='hash', group_aligned=False, n=1500):
    tot = 0.0
    for _ in range(n):
        # 64 consecutive records of one run: the run holds `presence` of the tile's rows, rows are uniform over the buckets' range
        span = 64 / presence / rows_per_tile                       # fraction of the tile's key range the round covers
        pos = np.sort(rng.uniform(0, span, 64)) + rng.uniform(0, 1 - span)
        b0 = (pos * (nb // 2)).astype(np.int64) * 2                  # ordered bucket (two slots = 16 bytes)
        lanes = np.arange(64)
        if group_aligned:
            order = np.array([l for g in G128 for l in g])          # record r sits in lane order[r]
            rec_of_lane = np.empty(64, dtype=np.int64); rec_of_lane[order] = lanes
            b0 = b0[rec_of_lane]
        h0 = rng.integers(0, nb // 2, 64) * 2
        h1 = rng.integers(0, nb // 2, 64) * 2
        if scheme == 'hash':
            c = cycles(h0 * 8, 16, G128, 64) + cycles(h1 * 8, 16, G128, 64); slot = h0 + rng.integers(0, 2, 64)
        elif scheme == 'ordered+hash':
            c = cycles(b0 * 8, 16, G128, 64) + cycles(h1 * 8, 16, G128, 64); slot = b0 + rng.integers(0, 2, 64)
        else:                                                        # one ordered bucket of four slots
            b4 = b0 // 4 * 4
            c = cycles(b4 * 8, 16, G128, 64) + cycles(b4 * 8 + 16, 16, G128, 64); slot = b4 + rng.integers(0, 4, 64)
        tot += c + cycles(slot * 4, 4, G32, 32)
    return tot / n
print('LDS cycles per round of 64 records (conflict-free: 4 + 4 + 2 = 10):')
for name, presence in (('configs[2] 20v20 (a run holds 65 % of the rows)', 0.65), ('4v4 (57 %)', 0.57), ('rows of 3 records of 40 samples (7 %)', 0.07)):
    r = {sch + (' aligned' if al else ''): round(round_cycles(0, presence=presence, scheme=sch, group_aligned=al), 1)
         for sch in ('hash', 'ordered+hash', 'ordered4') for al in (False, True) if not (sch == 'hash' and al)}
    print('  %-50s %s' % (name, r))
