"""The chain's event writer pulls (k_cpull, v2e_amd/csrc/emu_chain.h): output row j of an iteration holds the event of canonical index
sigma^-1(j), found by a search over the per-key group prefixes and a bit select in the group's pixel ballots.  This is a numpy model of
exactly that index logic -- k_ctot's ballots and totals, k_cframe's prefixes and row bases, k_cpull's per-row walk -- against the
DEFINITION of the order (emulator.py:861-870 as oracle/emu_oracle.c:504-535 restates it for Philox mode: per iteration the ON events in
pixel order, then the OFF events, placed at sigma(canonical index); shot-noise events behind all signal events, ON block then OFF block).
The device kernels are checked against the reference's digests on the GPU; this keeps the algorithm itself under the CPU suite."""
import numpy as np
import pytest

GROUP_PX = 256


def definition_rows(mag, neg, shot_on, shot_off, key, oracle_lib):
    """[(iteration or -1 for shot, pixel, polarity)] in output order, by pushing every event to sigma(its canonical index)."""
    seed, clip, frame = key
    rows = []
    for i in range(int(mag.max(initial=0))):
        cand = mag > i
        on = np.flatnonzero(cand & ~neg)
        off = np.flatnonzero(cand & neg)
        canon = [(i, int(p), +1) for p in on] + [(i, int(p), -1) for p in off]
        n = len(canon)
        if n == 0:
            continue
        inv = oracle_lib.perm_idx(seed, clip, frame, i, n)  # inv[sigma(c)] = c: the forward map, as the push writer applies it
        rows += [canon[int(c)] for c in inv]
    rows += [(-1, int(p), +1) for p in np.flatnonzero(shot_on)] + [(-1, int(p), -1) for p in np.flatnonzero(shot_off)]
    return rows


def pull_rows(mag, neg, shot_on, shot_off, key, oracle_lib, two_level):
    seed, clip, frame = key
    npx = mag.size
    ngroups = (npx + GROUP_PX - 1) // GROUP_PX
    nwp = (ngroups + 15) // 16 * 16
    pad = ngroups * GROUP_PX - npx
    M = int(mag.max(initial=0))
    nkeys = 2 + 2 * M

    def ballots(flags):  # [ngroups][4] uint64, bit = lane of the sub-group (k_ctot)
        f = np.concatenate([flags, np.zeros(pad, bool)]).reshape(ngroups, 4, 64)
        return np.packbits(f, axis=2, bitorder="little").view(np.uint64).reshape(ngroups, 4)

    masks = np.zeros((nkeys, nwp, 4), np.uint64)                 # key-major, as CEmitArgs::cmask
    masks[0, :ngroups], masks[1, :ngroups] = ballots(shot_on), ballots(shot_off)
    for i in range(M):
        masks[2 + 2 * i, :ngroups] = ballots((mag > i) & ~neg)
        masks[3 + 2 * i, :ngroups] = ballots((mag > i) & neg)
    tot = np.array([[sum(bin(int(w)).count("1") for w in masks[k, g]) for g in range(nwp)] for k in range(nkeys)], np.int64)
    pre = np.cumsum(tot, axis=1) - tot                             # exclusive prefix over groups per key (k_cframe*)
    T = tot.sum(axis=1)
    n_signal = int(T[2:].sum())
    kbase = np.concatenate([[0, 0], np.cumsum(T[2:]) - T[2:]])     # first row of a key ... of its ITERATION for the OFF key:
    for i in range(M):
        kbase[3 + 2 * i] = kbase[2 + 2 * i]
    rows = []

    def locate(k, cp):
        row = pre[k]
        if two_level:  # every 16th entry, then the line of 16
            coarse = row[::16]
            gc = int(np.searchsorted(coarse, cp, side="right")) - 1
            line = row[16 * gc:16 * gc + 16]
            g = 16 * gc + int(np.searchsorted(line, cp, side="right")) - 1
        else:
            g = int(np.searchsorted(row[:ngroups], cp, side="right")) - 1   # the LAST group whose prefix is <= cp
        assert tot[k, g] > 0 and pre[k, g] <= cp < pre[k, g] + tot[k, g]
        bit = oracle_lib.nth_set_bit_256(masks[k, g], cp - int(pre[k, g]))
        return g * GROUP_PX + bit

    for i in list(range(M)) + [M]:
        k0 = 2 + 2 * i if i < M else 0
        base = int(kbase[k0]) if i < M else n_signal
        T_on, T_off = int(T[k0]), int(T[k0 + 1])
        n = T_on + T_off
        if n == 0:
            continue
        inv = oracle_lib.perm_inv_idx(seed, clip, frame, i, n) if i < M else np.arange(n)   # v2e_perm_invert per row
        for jj in range(n):
            assert len(rows) == base + jj
            c = int(inv[jj])
            eneg = c >= T_on
            p = locate(k0 + (1 if eneg else 0), c - T_on if eneg else c)
            rows.append((i if i < M else -1, p, -1 if eneg else +1))
    return rows


@pytest.mark.parametrize("two_level", [False, True])
@pytest.mark.parametrize("shape", [(1, 7), (3, 100), (37, 91), (64, 96), (50, 401)])
def test_pull_index_logic_reproduces_the_defined_order(shape, two_level, oracle_lib):
    H, W = shape
    rng = np.random.default_rng(H * 1000 + W)
    for case in range(3):
        npx = H * W
        mag = rng.poisson([0.05, 0.6, 2.5][case], npx).astype(np.int64)
        if case == 2:
            mag[rng.integers(0, npx, 3)] = 40                         # a few pixels with many events: sparse high iterations
        if case == 0 and npx > 300:
            mag[:GROUP_PX] = 0                                         # an empty first group (prefix shared with the next one)
        neg = rng.random(npx) < 0.45
        shot_on = rng.random(npx) < 0.01
        shot_off = (rng.random(npx) < 0.01) & ~shot_on
        key = (12345 + case, 0, 77 + case)
        want = definition_rows(mag, neg, shot_on, shot_off, key, oracle_lib)
        got = pull_rows(mag, neg, shot_on, shot_off, key, oracle_lib, two_level)
        assert got == want, (shape, case)
