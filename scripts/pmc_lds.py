#!/usr/bin/env python
"""Summarise a `rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE` pass: per kernel, the share of LDS-array
cycles that are bank-conflict replays and the LDS array's busy share of the launch (cycles summed over the chip / (GRBM_GUI_ACTIVE x CUs)).

  python scripts/pmc_lds.py <counter_dir> <out.json> [n_cus]
"""
import collections
import csv
import glob
import json
import sys


def main():
    n_cus = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name']][r['Counter_Name']] += float(r['Counter_Value'])
            n[r['Kernel_Name']].add(r['Dispatch_Id'])
    out = {}
    for k, c in acc.items():
        idx = c.get('SQ_LDS_IDX_ACTIVE', 0.0)
        if idx <= 0:
            continue
        act = c.get('GRBM_GUI_ACTIVE', 0.0)
        out[k] = {'launches': len(n[k]), 'lds_idx_active': idx, 'lds_bank_conflict': c.get('SQ_LDS_BANK_CONFLICT', 0.0),
                  'conflict_share_of_lds_cycles': c.get('SQ_LDS_BANK_CONFLICT', 0.0) / idx,
                  'lds_busy_share_of_launch': idx / (act * n_cus) if act > 0 else None}
    json.dump({'source': 'rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE', 'kernels': out}, open(sys.argv[2], 'w'), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]['lds_idx_active']):
        print('%-90s %4d  conflicts %.3f of LDS cycles, LDS busy %s of the launch' % (k[:90], v['launches'], v['conflict_share_of_lds_cycles'],
              'n/a' if v['lds_busy_share_of_launch'] is None else '%.3f' % v['lds_busy_share_of_launch']))


if __name__ == '__main__':
    main()
