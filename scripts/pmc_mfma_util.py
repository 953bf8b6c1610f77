#!/usr/bin/env python
"""Summarise a `rocprofv3 --kernel-trace --pmc MfmaUtil GRBM_GUI_ACTIVE` pass of bench.py: per kernel, launches and the launch-time-weighted
MfmaUtil (rocprofv3's derived metric: SQ_VALU_MFMA_BUSY_CYCLES summed over the chip / (GRBM_GUI_ACTIVE x SIMDs), percent).

  python scripts/pmc_mfma_util.py <counter_dir> <out.json>
"""
import collections
import csv
import glob
import json
import sys


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name']][r['Counter_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    out = {}
    for k, c in acc.items():
        if 'MfmaUtil' not in c:
            continue
        util = dict(c['MfmaUtil'])
        act = dict(c.get('GRBM_GUI_ACTIVE', []))
        ids = sorted(util)
        w = [act.get(i, 1.0) for i in ids]
        tot = sum(w) or 1.0
        out[k] = {'launches': len(ids), 'mfma_util_percent_time_weighted': sum(util[i] * wi for i, wi in zip(ids, w)) / tot,
                  'mfma_util_percent_min': min(util.values()), 'mfma_util_percent_max': max(util.values())}
    json.dump({'source': 'rocprofv3 --kernel-trace --pmc MfmaUtil GRBM_GUI_ACTIVE -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline',
               'kernels': out}, open(sys.argv[2], 'w'), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]['launches']):
        if v['mfma_util_percent_max'] > 1.0:
            print('%-80s %4d  %.1f %% (%.1f .. %.1f)' % (k[:80], v['launches'], v['mfma_util_percent_time_weighted'], v['mfma_util_percent_min'], v['mfma_util_percent_max']))


if __name__ == '__main__':
    main()
