#!/usr/bin/env python3
"""Per-kernel HBM rate of a profiled configuration: PMC bytes per step (profiles/pmc_traffic_<cfg>.json) over the kernel's serial time per step
(profiles/<round>_<cfg>_serial_kernel_stats.csv, wgrad on the main stream: no co-running kernel stretches the durations).
usage: python3 profiles/bandwidth_table.py round5 bf16 [b256]   ->  stdout (committed as profiles/<round>_<cfg>_kernel_bandwidth.txt)"""
import csv
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def short(name):
    return re.sub(r'\(.*', '', name).replace('void ', '')


def main():
    rnd, cfg = sys.argv[1], sys.argv[2]
    tag = sys.argv[3] if len(sys.argv) > 3 else 'b256'
    traffic = json.load(open(os.path.join(HERE, 'pmc_traffic_%s.json' % cfg)))
    stats = os.path.join(HERE, '%s_%s_%s_serial_kernel_stats.csv' % (rnd, cfg, tag))
    rows = list(csv.DictReader(open(stats)))
    per = {short(r['Name']): (int(r['Calls']), float(r['TotalDurationNs'])) for r in rows}
    nsteps = [v[0] for k, v in per.items() if 'softmax' in k][0]
    excluded = set(traffic.get('excluded_setup_kernels', []))
    out = []
    for name, e in traffic['kernels'].items():
        s = short(name)
        if s not in per or name in excluded or 'FillFunctor' in name or name.startswith('__amd_rocclr'):
            continue
        calls, ns = per[s]
        ms = ns / 1e6 / nsteps
        gb = e['bytes_per_launch'] * e['launches_profiled'] / traffic['steps_profiled'] / 1e9
        out.append((ms, gb, gb / ms if ms else 0.0, calls / nsteps, s))
    out.sort(reverse=True)
    tot_ms, tot_gb = sum(o[0] for o in out), sum(o[1] for o in out)
    print('# %s %s: PMC bytes per step / serial kernel time per step (build %s, %d steps timed, %d steps counted)' % (rnd, cfg, traffic['build_id'][:12], nsteps,
                                                                                                                    traffic['steps_profiled']))
    print('# %8s %8s %7s %6s  kernel' % ('ms/step', 'GB/step', 'TB/s', 'n'))
    for ms, gb, bw, n, s in out:
        if ms < 0.02:
            continue
        print('  %8.3f %8.2f %7.2f %6.1f  %s' % (ms, gb, bw, n, s[:110]))
    print('# total %8.3f ms %8.2f GB  %.2f TB/s; at 6.3 TB/s the bytes alone take %.2f ms' % (tot_ms, tot_gb, tot_gb / tot_ms, tot_gb / 6.3))


if __name__ == '__main__':
    main()
