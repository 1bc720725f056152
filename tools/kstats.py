"""Print the top rows of a rocprofv3 kernel_stats.csv found under a directory."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 15]:
    print('%-72s %6s %12s %10s' % (r['Name'][:72], r['Calls'], r['TotalDurationNs'], r['AverageNs']))
