"""Flip-aware comparison of an fp32 device step with the float64 oracle (test infrastructure).

A pre-activation within fp32 rounding of zero can fall on the other side of a ReLU on the device (fp32 sums, its own summation order)
than in the oracle (float64).  ONE such decision moves the gradient of a small tensor by 1e-3 ... 1e-2 — far above the 1e-3 bar — although
nothing is wrong.  Instead of widening the bar for every tensor, the comparison here keeps it and accounts for the decisions:

  1. the oracle step records its near-ties (|z| <= tie_tol * rms(z) at a ReLU input; oracle.net.Tape.near_ties);
  2. if every gradient tensor is within `tol` of the device's, done;
  3. otherwise the effect of inverting each near-tie decision on every gradient tensor is measured (the oracle re-runs that tower with the
     decision inverted: oracle.net.Tape.relu_flips), and a greedy search picks the decisions whose inversion brings ALL tensors within
     `tol` — at most `max_flips` of them, each one an element the oracle itself marked as a tie.  No such set => the test fails;
  4. the oracle state advances with the chosen decisions, so later steps and the parameter comparison follow the device's branch.
"""
import numpy as np

from oracle import net as ON

# What the tests that use flip_aware_step allow per step, and what they have needed so far (rounds 3-4: 0-2).  The search itself may take up
# to `max_flips` = 6 inversions; a test asserts its own count against THIS budget with the count in the message (a drift
# from 1-2 to 6 must not pass unnoticed).
FLIP_BUDGET = 3


def assert_flip_budget(flips, where=''):
    """Every use reports the number of inverted near-tie decisions and fails above FLIP_BUDGET."""
    n = len(flips)
    assert n <= FLIP_BUDGET, '{}: {} ReLU near-tie decisions had to be inverted to meet the 1e-3 bar (budget {}; rounds 3-4 needed 0-2): {}'.format(
        where, n, FLIP_BUDGET, flips)
    return n


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _worst(dev, ref, keys):
    return max((rel_l2(dev[k], ref[k]), k) for k in keys)


def flip_aware_step(spec, state, towers, dev_grads, hp=None, batch_total=None, lr_mult=1.0, tol=1e-3, tie_tol=4e-6, max_flips=6, noise_floor=1e-9):
    """One oracle step on `towers` [(x, y), ...] from `state` (advanced in place) whose ReLU decisions at near-ties follow the device.
    dev_grads: the device's gradients of this step (tower mean, name -> array).  Returns (loss, pred, grads, flips) with `flips` the list
    of (tower, ReLU ordinal, flat index) decisions that were inverted (empty in the common case)."""
    s0 = ON.copy_state(state)
    ntow = len(towers)

    def run(st, flips=None, probe_ties=False):
        probe = {'relu_flips': {(t, k): idx for (t, k), idx in (flips or {}).items()}}
        if probe_ties:
            probe['tie_tol'] = tie_tol
        loss, pred, g = ON.train_step(spec, st, None, None, hp=hp, lr_mult=lr_mult, batch_total=batch_total, tower_batches=towers, probe=probe)
        return loss, pred, g, probe.get('near_ties', [])

    st = ON.copy_state(s0)
    loss, pred, g, ties = run(st, None, True)
    keys = [k for k in g if np.linalg.norm(g[k]) > noise_floor]
    worst = _worst(dev_grads, g, keys)
    if worst[0] <= tol:
        _assign(state, st)
        return loss, pred, g, []
    assert ties, 'gradient {} is {:.2e} away from the oracle and the oracle has no ReLU near-tie (|z| <= {:g} rms) to account for it'.format(worst[1], worst[0], tie_tol)
    assert len(ties) <= 64, 'too many near-ties ({}) for a flip search'.format(len(ties))
    # effect of inverting each near-tie decision on the tower-mean gradient: re-run that tower alone, with and without the inversion
    base = {}
    deltas = []
    for (ti, k, i) in ties:
        if ti not in base:
            base[ti] = ON.train_step(spec, ON.copy_state(s0), None, None, hp=hp, lr_mult=lr_mult, batch_total=batch_total, tower_batches=[towers[ti]])[2]
        gf = ON.train_step(spec, ON.copy_state(s0), None, None, hp=hp, lr_mult=lr_mult, batch_total=batch_total, tower_batches=[towers[ti]],
                           probe={'relu_flips': {(0, k): [i]}})[2]
        deltas.append({n: (gf[n] - base[ti][n]) / ntow for n in keys})
    resid = {n: np.asarray(dev_grads[n], np.float64) - g[n] for n in keys}
    norm = {n: max(np.linalg.norm(g[n]), 1e-30) for n in keys}

    def score(r):
        return sum((np.linalg.norm(r[n]) / norm[n]) ** 2 for n in keys)
    chosen = []
    for _ in range(max_flips):
        if max(np.linalg.norm(resid[n]) / norm[n] for n in keys) <= tol:
            break
        cur = score(resid)
        best = None
        for e, d in enumerate(deltas):
            if e in chosen:
                continue
            sc = score({n: resid[n] - d[n] for n in keys})
            if best is None or sc < best[0]:
                best = (sc, e)
        if best is None or best[0] >= cur:
            break
        chosen.append(best[1])
        resid = {n: resid[n] - deltas[best[1]][n] for n in keys}
    flips = {}
    for e in chosen:
        ti, k, i = ties[e]
        flips.setdefault((ti, k), []).append(i)
    st = ON.copy_state(s0)
    loss, pred, g, _ = run(st, flips)
    worst = _worst(dev_grads, g, keys)
    assert worst[0] <= tol, ('gradient {} stays {:.2e} away from the oracle with {} of its {} ReLU near-tie decisions inverted'
                             .format(worst[1], worst[0], len(chosen), len(ties)))
    _assign(state, st)
    return loss, pred, g, [ties[e] for e in chosen]


def _assign(dst, src):
    dst.params, dst.stats, dst.accum, dst.ema, dst.ema_stats, dst.step = src.params, src.stats, src.accum, src.ema, src.ema_stats, src.step
