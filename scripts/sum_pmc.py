import csv, glob, sys, collections
for d in sys.argv[1:]:
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:60]
            acc[(k, r['Counter_Name'])][0] += 1
            acc[(k, r['Counter_Name'])][1] += float(r['Counter_Value'])
    for (k, c), (n, v) in sorted(acc.items()):
        if 'conv' in k or 'wgrad' in k:
            print(d, k, c, n, 'avg %.1f' % (v / n))
