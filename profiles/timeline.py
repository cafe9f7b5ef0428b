#!/usr/bin/env python3
"""Where does the step's wall time go when two HIP streams overlap?  Reads a rocprofv3 `--kernel-trace` CSV of bench.py, cuts out the
timed steps (a step starts at its `input_prep_kernel`), classifies every kernel as MFMA-class (conv / Winograd GEMMs) or HBM-class
(everything else) and integrates the timeline:

    only_mfma   wall time with >= 1 MFMA-class kernel and no HBM-class kernel in flight
    only_hbm    wall time with >= 1 HBM-class kernel and no MFMA-class kernel in flight   <- exposed bandwidth-bound passes
    both        both classes in flight (the overlap the second stream buys)
    idle        nothing in flight (launch gaps, event waits)

and lists, per kernel symbol, how much of its duration ran exposed (no kernel of the other class beside it).  The forward / backward
split is taken at the softmax kernel.  Usage: python3 profiles/timeline.py <kernel_trace.csv> [--steps K] [--json]
"""
import csv
import json
import re
import sys
from collections import defaultdict

MFMA = re.compile(r'conv_gemm_|conv_wino_|skinny_conv|fc_')


def short(name):
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*$', '', name)
    return name[:70]


def load(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], int(r.get('Queue_Id', 0) or 0)))
    rows.sort()
    return rows


def integrate(rows, t0, t1):
    """sweep over [t0, t1): time per state + exposed time per kernel symbol"""
    ev = []
    for i, (s, e, name, q) in enumerate(rows):
        s, e = max(s, t0), min(e, t1)
        if e <= s:
            continue
        ev.append((s, 1, i))
        ev.append((e, 0, i))
    ev.sort()
    live = set()
    state = defaultdict(int)
    exposed = defaultdict(int)
    total = defaultdict(int)
    last = t0
    for t, kind, i in ev:
        dt = t - last
        if dt > 0:
            nm = sum(1 for j in live if MFMA.search(rows[j][2]))
            nh = len(live) - nm
            key = 'idle' if not live else 'both' if (nm and nh) else 'only_mfma' if nm else 'only_hbm'
            state[key] += dt
            for j in live:
                total[short(rows[j][2])] += dt
                if (nm and not nh) or (nh and not nm):
                    if len(live) == 1 or all(bool(MFMA.search(rows[k][2])) == bool(MFMA.search(rows[j][2])) for k in live):
                        exposed[short(rows[j][2])] += dt
        last = t
        if kind:
            live.add(i)
        else:
            live.discard(i)
    if t1 > last:
        state['idle'] += t1 - last
    return state, exposed, total


def main():
    path = sys.argv[1]
    want = int(sys.argv[sys.argv.index('--steps') + 1]) if '--steps' in sys.argv else 3
    rows = load(path)
    starts = [s for s, e, n, q in rows if 'input_prep_kernel' in n]
    if len(starts) < want + 1:
        raise SystemExit('need at least {} steps in the trace (found {} input_prep launches)'.format(want + 1, len(starts)))
    starts = starts[-(want + 1):]                  # the last `want` complete steps (the trace ends with bench.py's own tail)
    out = {'steps': want, 'ms_per_step': (starts[-1] - starts[0]) / want / 1e6}
    for phase in ('step', 'forward', 'backward'):
        acc = defaultdict(int)
        exp = defaultdict(int)
        tot = defaultdict(int)
        for a, b in zip(starts[:-1], starts[1:]):
            sm = [s for s, e, n, q in rows if a <= s < b and 'softmax_xent' in n]
            mid = sm[0] if sm else (a + b) // 2
            lo, hi = {'step': (a, b), 'forward': (a, mid), 'backward': (mid, b)}[phase]
            st, ex, to = integrate(rows, lo, hi)
            for k, v in st.items():
                acc[k] += v
            for k, v in ex.items():
                exp[k] += v
            for k, v in to.items():
                tot[k] += v
        out[phase] = {k: round(v / want / 1e6, 3) for k, v in sorted(acc.items())}
        out[phase]['wall'] = round(sum(acc.values()) / want / 1e6, 3)
        top = sorted(exp.items(), key=lambda kv: -kv[1])[:14]
        out[phase]['exposed_ms_by_kernel'] = {k: [round(v / want / 1e6, 3), round(tot[k] / want / 1e6, 3)] for k, v in top}
    if '--json' in sys.argv:
        print(json.dumps(out))
        return
    print('ms/step {:.3f} over {} steps'.format(out['ms_per_step'], want))
    for phase in ('step', 'forward', 'backward'):
        p = out[phase]
        print('{:9s} wall {:7.3f}  only_mfma {:7.3f}  only_hbm {:7.3f}  both {:7.3f}  idle {:7.3f}'.format(
            phase, p['wall'], p.get('only_mfma', 0), p.get('only_hbm', 0), p.get('both', 0), p.get('idle', 0)))
        for k, (e, t) in p['exposed_ms_by_kernel'].items():
            print('    {:72s} exposed {:7.3f} of {:7.3f} ms'.format(k, e, t))


if __name__ == '__main__':
    main()
