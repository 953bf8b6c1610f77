#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter CSVs (separate FETCH_SIZE and WRITE_SIZE passes of the same bench.py command) into
profiles/<name>.json: per kernel, launches and average HBM-side bytes per launch.

MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2
of the bytes of a wide (16 B/lane) coalesced read stream, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact for
16-B-per-lane stores.  Both count at the L2's fabric side, so Infinity-Cache hits are included (upper bound on HBM bytes).

  python scripts/pmc_traffic.py <fetch_dir> <write_dir> <out.json>
"""
import collections
import csv
import glob
import json
import sys


def collect(d, counter):
    out = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        per = collections.defaultdict(float)
        name = {}
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                per[r['Dispatch_Id']] += float(r['Counter_Value'])
                name[r['Dispatch_Id']] = r['Kernel_Name']
        for k, v in per.items():
            out[name[k]][0] += 1
            out[name[k]][1] += v
    return out


def main():
    fetch = collect(sys.argv[1], 'FETCH_SIZE')
    write = collect(sys.argv[2], 'WRITE_SIZE')
    res = {}
    for k in sorted(set(fetch) | set(write)):
        nf, vf = fetch.get(k, [0, 0.0])
        nw, vw = write.get(k, [0, 0.0])
        res[k] = {'launches': max(nf, nw),
                  'read_bytes_per_launch': 2.0 * 1024.0 * vf / max(nf, 1),       # gfx950 correction: FETCH_SIZE counts 64 B per 128-B request
                  'write_bytes_per_launch': 1024.0 * vw / max(nw, 1)}
        res[k]['hbm_bytes_per_launch'] = res[k]['read_bytes_per_launch'] + res[k]['write_bytes_per_launch']
    json.dump({'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes) of bench.py', 'kernels': res}, open(sys.argv[3], 'w'), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches'])[:12]:
        print('%-70s n=%4d  %.1f MB/launch' % (k[:70], v['launches'], v['hbm_bytes_per_launch'] / 1e6))


if __name__ == '__main__':
    main()
