"""Per-kernel sums of rocprofv3 --pmc counters: python tools/pmc_kernel_table.py <dir> [name filter]
(reads every *counter_collection.csv under <dir>; prints counter totals per kernel name and per launch)."""
import collections
import csv
import glob
import os
import re
import sys


def main():
    root = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name']
            if flt and flt not in n:
                continue
            n = re.sub(r'\(anonymous namespace\)::', '', n)
            n = re.sub(r'\((?!anonymous)[^()]*(\([^()]*\)[^()]*)*\)( const)?$', '', n)[:90]
            tot[n][r['Counter_Name']] += float(r['Counter_Value'])
            calls[(n, r['Counter_Name'])].add(r['Dispatch_Id'])
    for n, cs in sorted(tot.items()):
        print(n)
        for c, v in sorted(cs.items()):
            k = max(len(calls[(n, c)]), 1)
            print('    %-28s %16.0f   per launch %14.0f   (%d launches)' % (c, v, v / k, k))


if __name__ == '__main__':
    main()
