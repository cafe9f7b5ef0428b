"""A bench line must be self-consistent: every derived figure (`frac`, `gbs`, `hbm_frac`, `e2e_mfma_frac`, `value` against `ms_per_step` ...) recomputed
from the line's OWN fields (bench.verify_line, which main() also asserts before printing).  Runs over the lines committed under profiles/ for this
round and later; the older lines carry the Winograd launch-count inconsistency VERDICT r4 named and are only used as test material here."""
import glob
import json
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _lines(min_round):
    out = []
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'round*_bench.json'))):
        m = re.match(r'round(\d+)_', os.path.basename(f))
        if m and int(m.group(1)) >= min_round:
            out.append(f)
    return out


def test_committed_bench_lines_are_self_consistent():
    import bench
    files = _lines(5)
    assert files, 'no bench line of round >= 5 under profiles/ (profiles/collect.sh writes them)'
    for f in files:
        with open(f) as fh:
            line = json.load(fh)
        assert bench.verify_line(line) == [], f


def test_verify_line_sees_an_inconsistent_line():
    import bench
    f = os.path.join(ROOT, 'profiles', 'round4_default_bench.json')
    with open(f) as fh:
        line = json.load(fh)
    # the round-4 line: consistent except for the Winograd entries, which do not say how many tail launches their brackets hold
    bad = bench.verify_line(line)
    assert bad and all('Winograd' in b for b in bad), bad
    for k, e in line['roofline_by_kernel'].items():
        if k.startswith('conv_wino'):
            e['tail_launches_per_step'] = 0
    assert bench.verify_line(line) == []
    line['roofline']['frac'] += 0.01
    assert any(b.startswith('roofline.frac') for b in bench.verify_line(line))
    line['roofline']['frac'] -= 0.01
    k = next(iter(line['roofline_by_kernel']))
    line['roofline_by_kernel'][k]['gbs'] *= 1.5
    assert any('gbs' in b for b in bench.verify_line(line))
    line['roofline_by_kernel'][k]['gbs'] /= 1.5
    line['value'] *= 1.02
    assert any(b.startswith('value') for b in bench.verify_line(line))


def test_pmc_summaries_count_their_steps():
    """profiles/pmc_traffic_*.json: steps_profiled is the number of softmax launches in the profiled run (not a constant), and the per-step total leaves
    the allocation-time fills out"""
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'pmc_traffic_*.json'))):
        with open(f) as fh:
            d = json.load(fh)
        if 'bytes_per_step_total' not in d:
            pytest.skip('{} predates round 5 (re-run profiles/collect.sh)'.format(os.path.basename(f)))
        once = sum(v['launches_profiled'] for k, v in d['kernels'].items() if k.startswith('softmax_xent'))
        assert once == d['steps_profiled'], f
        tot = sum(v['bytes_per_launch'] * v['launches_profiled'] for k, v in d['kernels'].items() if k not in d['excluded_setup_kernels']) / d['steps_profiled']
        assert abs(tot - d['bytes_per_step_total']) <= 1e-6 * tot, f
        assert all('FillFunctor' in k or k.startswith('__amd_rocclr') for k in d['excluded_setup_kernels'])
