"""Averages the counters of the rocprofv3 --pmc passes over profiles/probes/pmc_layers.py per (kernel symbol, grid size): usage
python3 profiles/probes/pmc_layers_summary.py <dir with the passes' *_counter_collection.csv files> > profiles/roundN_pmc_layers_bf16.txt"""
import collections
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from summarize import short  # noqa: E402


def main(root):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True)):
        for r in csv.DictReader(open(f)):
            name = short(r['Kernel_Name'])
            if not name.startswith('conv_gemm'):
                continue
            key = (name, int(r['Grid_Size']) // int(r['Workgroup_Size']))
            acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
            dur[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    counters = sorted({c for v in acc.values() for c in v})
    print('# per launch (mean over the profiled launches; durations include the counter collection overhead); grid = workgroups')
    print('%-52s %8s %9s ' % ('kernel', 'grid', 'us') + ' '.join('%22s' % c for c in counters))
    for key in sorted(acc):
        v = acc[key]
        print('%-52s %8d %9.1f ' % (key[0][:52], key[1], sum(dur[key]) / len(dur[key])) + ' '.join('%22.4g' % (sum(v[c]) / len(v[c])) if c in v else '%22s' % '-' for c in counters))


if __name__ == '__main__':
    main(sys.argv[1])
