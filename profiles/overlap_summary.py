"""Summarise the 2-rank rocprofv3 traces of profiles/trace_2rank.sh: how much of the gradient all-reduce runs beside the
backward kernels.  usage: overlap_summary.py <gpurun_out/trace2> <round tag>

Per rank and per training step (a step ends with the SGD update kernel): every blit of a gradient bucket (gloo reduces on the
host: device -> host blit, host all-reduce, host -> device blit, on gloo's own streams) with the share of its duration during
which a compute kernel of the same process was executing, the host-reduce windows between the two blits of a bucket and the
same share for them, when the first bucket left relative to the start of the step, and the exposed tail = last blit end minus
last compute kernel end (what the collective adds behind the backward pass).  Writes profiles/<round>_overlap_2rank.json."""
import csv, glob, json, os, sys


def load(pattern):
    rows = []
    for f in glob.glob(pattern, recursive=True):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    return rows


def covered(iv, busy):
    """length of interval iv = (a, b) covered by the union of the sorted, merged intervals in busy"""
    a, b = iv
    tot = 0
    for s, e in busy:
        if e <= a:
            continue
        if s >= b:
            break
        tot += min(b, e) - max(a, s)
    return tot


def merge(ivs):
    out = []
    for s, e in sorted(ivs):
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def rank_summary(d):
    ks = load(os.path.join(d, '**', '*kernel_trace.csv'))
    ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id', '')) for r in ks)
    upd = [k for k in ks if 'sgd_' in k[2]]
    if not upd:
        return {'error': 'no update kernel in the trace'}
    # a step ends with its last update kernel (the update is 1-2 launches a few microseconds apart)
    bounds = [u[1] for i, u in enumerate(upd) if i + 1 == len(upd) or upd[i + 1][0] - u[1] > 5_000_000]
    # the bucket copies of torch's gloo backend are blit KERNELS on its own streams (device -> pinned host, host all-reduce,
    # pinned host -> device); the memory-copy trace only holds the small host-initiated copies
    is_copy = lambda k: 'copyBuffer' in k[2]
    steps = []
    for i in range(1, len(bounds)):
        lo, hi = bounds[i - 1], bounds[i]
        kk = [k for k in ks if lo <= k[0] < hi]
        comp = [k for k in kk if not is_copy(k) and 'sgd_' not in k[2]]
        cc = [k for k in kk if is_copy(k) and k[1] - k[0] >= 50_000]                # gradient buckets (>= 50 us blits)
        if not comp:
            continue
        busy = merge([(k[0], k[1]) for k in comp])
        cdur = sum(c[1] - c[0] for c in cc)
        cov = sum(covered((c[0], c[1]), busy) for c in cc)
        # per gloo stream: first blit = D2H, second = H2D; the window between them is the host all-reduce of that bucket
        by_stream = {}
        for c in cc:
            by_stream.setdefault(c[3], []).append(c)
        win = [(v[j][1], v[j + 1][0]) for v in by_stream.values() for j in range(0, len(v) - 1, 2) if v[j + 1][0] > v[j][1]]
        wdur = sum(b - a for a, b in win)
        wcov = sum(covered(w, busy) for w in win)
        last_comp = max(k[1] for k in comp)
        last_copy = max(c[1] for c in cc) if cc else last_comp
        steps.append({'step_ms': (hi - lo) / 1e6, 'compute_busy_ms': sum(e - s for s, e in busy) / 1e6, 'bucket_blits': len(cc),
                      'blit_ms': cdur / 1e6, 'blit_under_compute': cov / cdur if cdur else None,
                      'host_reduce_windows': len(win), 'host_reduce_ms': wdur / 1e6, 'host_reduce_under_compute': wcov / wdur if wdur else None,
                      'first_blit_after_step_start_ms': (min(c[0] for c in cc) - lo) / 1e6 if cc else None,
                      'exposed_tail_ms': max(0, last_copy - last_comp) / 1e6})
    return {'steps': steps}


def main():
    root, tag = sys.argv[1], sys.argv[2]
    out = {'what': 'bench.py --gpus 2 on one MI355X (both ranks on GPU 0, gloo all-reduce through the host), rocprofv3 --kernel-trace --memory-copy-trace per rank',
           'ranks': {r: rank_summary(os.path.join(root, r)) for r in ('r0', 'r1')}}
    for r in ('r0', 'r1'):
        try:
            out['ranks'][r]['bench'] = json.loads(open(os.path.join(root, r + '.json')).read().strip().splitlines()[-1])
        except Exception:
            pass
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, '{}_overlap_2rank.json'.format(tag)), 'w') as f:
        json.dump(out, f, indent=1)
    for r, v in out['ranks'].items():
        for s in v.get('steps', [])[-4:]:
            print(r, {k: (round(x, 3) if isinstance(x, float) else x) for k, x in s.items()})


if __name__ == '__main__':
    main()
