"""Turn the raw rocprofv3 output of profiles/collect.sh into the small committed summaries:
  <round>_<dtype>_b256_kernel_stats.csv   rocprofv3's own per-kernel --stats table
  <round>_<dtype>_b256_bench.json         the JSON line bench.py printed under the profiler
  pmc_traffic_<dtype>.json               HBM bytes per launch per kernel = (2*FETCH_SIZE + WRITE_SIZE) * 1024
(FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes for wide
coalesced reads, so it is doubled — MI355X_MICROARCH.md, "HBM".)"""
import csv
import glob
import re
import json
import os
import shutil
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def demangle(name):
    """rocprofv3's demangler gives up on __bf16 template arguments (DF16b); decode the few forms our kernels use."""
    m = re.match(r'_Z(\d+)', name)
    if not m:
        return name
    n = int(m.group(1))
    base, rest = name[m.end():m.end() + n], name[m.end() + n:]
    if not rest.startswith('I'):
        return base
    args, i = [], 1
    while i < len(rest) and rest[i] != 'E':
        if rest.startswith('DF16b', i):
            args.append('bf16')
            i += 5
        elif rest[i] == 'f':
            args.append('float')
            i += 1
        elif rest[i] == 'L':
            j = rest.index('E', i)
            kind, val = rest[i + 1], rest[i + 2:j]
            args.append(('true' if val == '1' else 'false') if kind == 'b' else val)
            i = j + 1
        else:
            return base + '<?>'
    return base + '<' + ', '.join(args) + '>'


def short(name):
    name = name.replace('void ', '')
    if name.startswith('_Z'):
        return demangle(name)
    i = name.find('(')
    return name[:i] if i > 0 else name


def counter_rows(d):
    f = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    if not f:
        return {}
    acc = defaultdict(lambda: [0, 0.0])
    with open(f[0]) as fh:
        for r in csv.DictReader(fh):
            a = acc[short(r['Kernel_Name'])]
            a[0] += 1
            a[1] += float(r['Counter_Value'])
    return acc


def build_id():
    """digest of the kernel sources of THIS tree (= mcn_build_id() of the library the profiled run loaded: _ffi refuses a stale one);
    bench.py reports `traffic: null` when the committed summary was taken from another build"""
    import importlib.util
    spec = importlib.util.spec_from_file_location('_mcn_build', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'myconvnet_amd', 'build.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.source_digest()


def main(root, rnd='round2'):
    here = os.path.dirname(os.path.abspath(__file__))
    # (tag of the raw directories / output files, --dtype, batch, PMC command)
    configs = [('fp32', 'fp32', 256, 'python3 bench.py --steps 1 --warmup 1 --dtype fp32 --no-secondary --no-cpu-baseline'),
               ('bf16', 'bf16', 256, 'python3 bench.py --steps 1 --warmup 1 --dtype bf16 --no-secondary --no-cpu-baseline'),
               ('efficientnet_b0_bf16', 'bf16', 512, 'python3 bench.py --model efficientnet_b0 --dtype bf16 --steps 1 --warmup 1'),
               ('deeplabv3plus_bf16', 'bf16', 16, 'python3 bench.py --model deeplabv3plus --dtype bf16 --steps 1 --warmup 1')]
    for dt, dtype, batch, cmd in configs:
        for src, tag in (('stats_', ''), ('serial_', 'serial_')):
            st = glob.glob(os.path.join(root, src + dt, '**', '*kernel_stats.csv'), recursive=True)
            if st:
                with open(st[0]) as fh, open(os.path.join(here, '%s_%s_b%d_%skernel_stats.csv' % (rnd, dt, batch, tag)), 'w') as out:
                    rd = csv.reader(fh)
                    wr = csv.writer(out, quoting=csv.QUOTE_MINIMAL)
                    for i, row in enumerate(rd):
                        if i and row:
                            row[0] = short(row[0]) if row[0].startswith('_Z') else row[0]
                        wr.writerow(row)
        bench_line = None
        for src, tag in (('bench_', ''), ('serial_', 'serial_')):
            bj = os.path.join(root, '%s%s.json' % (src, dt))
            if os.path.exists(bj) and os.path.getsize(bj):
                shutil.copy(bj, os.path.join(here, '%s_%s_b%d_%sbench.json' % (rnd, dt, batch, tag)))
                try:
                    bench_line = json.load(open(bj))
                except ValueError:
                    pass
        fe = counter_rows(os.path.join(root, 'pmc_FETCH_SIZE_' + dt))
        wr = counter_rows(os.path.join(root, 'pmc_WRITE_SIZE_' + dt))
        if not fe or not wr:
            continue
        kernels = {}
        for k in fe:
            if k not in wr or fe[k][0] != wr[k][0]:
                continue
            n = fe[k][0]
            fetch = fe[k][1] / n * 1024.0
            write = wr[k][1] / n * 1024.0
            kernels[k] = {'launches_profiled': n, 'fetch_size_bytes_raw': round(fetch), 'write_size_bytes': round(write),
                          'bytes_per_launch': round(2.0 * fetch + write)}
        # steps in the profiled run: counted, not assumed (1 warm-up + 1 timed step + the repetitions of bench.py's instrumented pass; VERDICT r4: the
        # constant 5 was wrong for a run with the extra fetch_true steps) — a training step launches exactly one softmax cross-entropy kernel
        once = [v['launches_profiled'] for k, v in kernels.items() if k.startswith('softmax_xent')]
        steps = sum(once) if once else 5
        # per-step HBM bytes of the STEP's kernels: allocation-time fills (torch.zeros of every activation buffer, 23-48 GB, once per process) and
        # runtime copies are set-up, not step traffic
        setup = sorted(k for k in kernels if 'FillFunctor' in k or k.startswith('__amd_rocclr'))
        step_bytes = sum(v['bytes_per_launch'] * v['launches_profiled'] for k, v in kernels.items() if k not in setup) / max(steps, 1)
        out = {'batch': batch, 'dtype': dtype, 'build_id': build_id(), 'command': cmd,
               'steps_profiled': steps, 'bytes_per_step_total': round(step_bytes), 'excluded_setup_kernels': setup,
               'formula': 'bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024, averaged over every launch of the kernel in the run; '
                          'bytes_per_step_total = sum over the step\'s kernels (set-up fills / copies excluded) of bytes_per_launch * launches_profiled / steps_profiled',
               'kernels': kernels}
        if bench_line and 'calls_per_step' in bench_line:
            out['calls_per_step'] = bench_line['calls_per_step']          # C-ABI calls per step (bench.py sums a call's kernels: CALL_KERNELS)
        with open(os.path.join(here, 'pmc_traffic_%s.json' % dt), 'w') as fh:
            json.dump(out, fh, indent=1, sort_keys=True)
        top = sorted(kernels.items(), key=lambda kv: -kv[1]['bytes_per_launch'] * kv[1]['launches_profiled'])[:6]
        for k, v in top:
            print(dt, k, v)


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof', sys.argv[2] if len(sys.argv) > 2 else 'round2')
