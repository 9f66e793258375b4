"""Design check of the k_chain speculation / redo protocol (v2e_amd/csrc/emu_chain.h) on a toy pixel model, on the CPU.

The HIP kernel is pinned event for event by the GPU fixture tests; what those cannot do is throw thousands of random
rule-on patterns at the CONTROL FLOW.  This file restates that control flow -- K frames per launch speculated "rule off",
the rule-on maxima a launch publishes, the next launch's check of that row, redo passes that take the first disagreeing
frame exactly and run the later frames under the row's values as predictions, mid-launch checkpoints every CHAIN_SUB
frames, ping-pong state planes, the tail launch, the lock-step frames of a redo pass (round 4) -- around a toy pixel whose update depends on the frame's global max
count M exactly when the rule is on, and compares it with plain frame-by-frame processing.
"""
import numpy as np
import pytest

CHAIN_SUB = 8


class Toy:
    """A pixel model with the dependency structure of the DVS pixel: counts from (input - base), M = max count over all
    pixels, a 'refractory' filter that is applied iff M >= mon and whose outcome depends on M and on a per-pixel
    timestamp, then base / timestamp updates."""

    def __init__(self, npx, mon, seed):
        rng = np.random.default_rng(seed)
        self.npx, self.mon = npx, mon
        self.base0 = rng.integers(0, 50, npx).astype(np.int64)
        self.ts0 = np.zeros(npx, np.int64)

    @staticmethod
    def count(base, x):
        return np.maximum((x - base) // 7, 0)

    def finalise(self, base, ts, c, f, m_exact):
        """m_exact: the frame's M if the rule is (taken to be) on, else 0.  Returns (base, ts, passed)."""
        if m_exact == 0:
            passed = c.copy()
            ts2 = np.where(c > 0, 1000 * (f + 1), ts)
        else:
            # event i of the frame happens at 1000 f + (i + 1) * 1000 // M; it passes if it is > 300 after the last one
            passed = np.zeros_like(c)
            ts2 = ts.copy()
            for i in range(int(c.max()) if c.size else 0):
                t = 1000 * f + (i + 1) * 1000 // m_exact
                ok = (c > i) & (t - ts2 > 300)
                passed += ok
                ts2 = np.where(ok, t, ts2)
        return base + 7 * passed, ts2, passed


def sequential(toy, frames):
    base, ts = toy.base0.copy(), toy.ts0.copy()
    out = []
    for f, x in enumerate(frames):
        c = toy.count(base, x)
        m = int(c.max())
        base, ts, passed = toy.finalise(base, ts, c, f, m if m >= toy.mon else 0)
        out.append((m, passed.copy()))
    return base, ts, out


def chain(toy, frames, K, n_groups=4, checkpoints=True, predict=True, lockstep=True):
    """The launch sequence of enqueue_run_chain + the pass loop of k_chain, all 'workgroups' of a launch in lock step."""
    F = len(frames)
    nB = (F + K - 1) // K
    nL = nB + 1                                   # + the tail launch that validates the last K frames
    groups = np.array_split(np.arange(toy.npx), n_groups)
    planes = [(toy.base0.copy(), toy.ts0.copy()), (toy.base0.copy(), toy.ts0.copy())]  # ping-pong: launch L reads [L%2]
    ck = [[None] * 4, [None] * 4]                 # [launch parity][checkpoint index] -> (base, ts)
    gM = np.zeros((nL, K + 1, K), np.int64)       # [launch][row = after r redo passes][frame of the launch]
    out = [None] * F                              # per frame: (wave maxima -> M, passed counts) as the last pass left them
    passes_total = 0

    def run_pass(f0, nf, c0, base, ts, exact, row_dst, last_exact, ck_set, lock_k=-1, pred=None):
        """Frames f0+c0 .. f0+nf-1 with exact[k] (0: rule off) as the M they are finalised under; publishes into row_dst.
        lock_k: the frame taken in lock-step (a redo pass, the frame behind one it finalised by the rule): its published
        maximum is read back after a rendezvous and IS its M; the lock-step goes on while the frames keep being rule-on.
        Returns (base, ts, last_exact)."""
        nonlocal passes_total
        passes_total += 1
        for k in range(c0, nf):
            if checkpoints and k > c0 and k % CHAIN_SUB == 0:
                ck_set[k // CHAIN_SUB - 1] = (base.copy(), ts.copy())
            f = f0 + k
            c = toy.count(base, frames[f])
            for g in groups:                      # every workgroup publishes its maximum iff it reaches the threshold
                wm = int(c[g].max()) if g.size else 0
                if k > last_exact and wm >= toy.mon:
                    row_dst[k] = max(row_dst[k], wm)
            if k == lock_k:                       # rendezvous, then every workgroup reads the row entry
                mk = int(row_dst[k])
                exact[k] = mk
                pred[k] = mk
                last_exact = k
                lock_k = k + 1 if (mk != 0 and k + 1 < nf) else -1
            base, ts, passed = toy.finalise(base, ts, c, f, int(exact[k]))
            out[f] = (int(c.max()), passed.copy())
        return base, ts, last_exact

    for L in range(nL):
        tail = L >= nB
        f0, nf = (F, 0) if tail else (L * K, min((L + 1) * K, F) - L * K)
        pf0, pnf = (L - 1) * K, (min(L * K, F) - (L - 1) * K) if L > 0 else 0
        base, ts = planes[L % 2][0].copy(), planes[L % 2][1].copy()
        redone = False
        if pnf > 0:
            pin = planes[(L + 1) % 2]             # what the previous launch started from
            rnd, last_exact = 0, -1
            exact = np.zeros(K, np.int64)
            pred = np.zeros(K, np.int64)
            while True:
                row = gM[L - 1][rnd]
                mism = [k for k in range(pnf) if k > last_exact and row[k] != pred[k]]
                if not mism:
                    break
                j = mism[0]
                for k in range(last_exact + 1, pnf):
                    exact[k] = row[k]             # < j verified, j exact, > j predictions ...
                    if not predict and k > j:
                        exact[k] = 0              # ... or, without prediction, speculate "off" again
                pred = np.where(np.arange(K) > j, exact, 0)
                last_exact = j
                rnd += 1
                redone = True
                c0 = (j // CHAIN_SUB) * CHAIN_SUB if checkpoints else 0
                if c0 == 0:
                    b, t = pin[0].copy(), pin[1].copy()
                else:
                    b, t = ck[(L + 1) % 2][c0 // CHAIN_SUB - 1]
                    b, t = b.copy(), t.copy()
                lock_k = j + 1 if (lockstep and row[j] != 0 and j + 1 < pnf) else -1
                base, ts, last_exact = run_pass(pf0, pnf, c0, b, t, exact, gM[L - 1][rnd], last_exact, ck[(L + 1) % 2], lock_k, pred)
                assert rnd <= K
            if redone:
                planes[L % 2] = (base.copy(), ts.copy())  # *_fix: the corrected input state of this launch
        if nf > 0:
            base, ts, _ = run_pass(f0, nf, 0, base, ts, np.zeros(K, np.int64), gM[L][0], -1, ck[L % 2])
            planes[(L + 1) % 2] = (base.copy(), ts.copy())
        else:
            planes[(L + 1) % 2] = (base.copy(), ts.copy())
    final = planes[nL % 2]
    return final[0], final[1], out, passes_total


@pytest.mark.parametrize("K", [1, 2, 3, 5, 8, 11, 16, 32])
@pytest.mark.parametrize("mon", [2, 4, 7])
def test_chain_protocol_equals_sequential(K, mon):
    for seed in range(6):
        rng = np.random.default_rng(100 * K + 10 * mon + seed)
        npx, F = 37, int(rng.integers(1, 75))
        toy = Toy(npx, mon, seed)
        drift = np.cumsum(rng.integers(0, 12, (F, 1)), axis=0)
        frames = (drift + rng.integers(0, 30, (F, npx))).astype(np.int64)
        b_ref, t_ref, out_ref = sequential(toy, frames)
        for ckp in (True, False):
            for pr in (True, False):
                for ls in (True, False):
                    b, t, out, _ = chain(toy, frames, K, checkpoints=ckp, predict=pr, lockstep=ls)
                    assert np.array_equal(b, b_ref) and np.array_equal(t, t_ref), (K, mon, seed, ckp, pr, ls)
                    for f in range(F):
                        assert out[f][0] == out_ref[f][0] and np.array_equal(out[f][1], out_ref[f][1]), (K, mon, seed, f, ls)


def test_prediction_and_checkpoints_save_passes():
    """On a clip whose rule-on frames cluster, predicting the later frames of a redo pass removes passes, and the
    protocol without either feature still gives the same result (what the A/B switches of the product select)."""
    rng = np.random.default_rng(7)
    npx, F, K = 64, 96, 32
    toy = Toy(npx, 4, 3)
    drift = np.cumsum(rng.integers(4, 14, (F, 1)), axis=0)
    frames = (drift + rng.integers(0, 40, (F, npx))).astype(np.int64)
    _, _, ref, _ = chain(toy, frames, K, checkpoints=False, predict=False, lockstep=False)
    n_plain = chain(toy, frames, K, checkpoints=False, predict=False, lockstep=False)[3]
    n_pred = chain(toy, frames, K, checkpoints=True, predict=True, lockstep=False)[3]
    n_lock = chain(toy, frames, K, checkpoints=True, predict=True, lockstep=True)[3]
    assert n_lock <= n_pred  # runs of rule-on frames: a rendezvous per frame of the run instead of a pass
    rule_on = sum(1 for m, _ in ref if m >= toy.mon)
    assert rule_on > 10
    assert n_pred < n_plain
