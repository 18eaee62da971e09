#!/usr/bin/env python3
"""Average kernel durations from a rocprofv3 kernel_stats csv (short names)."""
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    nm = re.sub(r'\(.*', '', r['Name']).replace('void ', '').replace('kvx::', '')
    if float(r['Percentage']) > 0.5:
        print("%-26s calls %5s avg %9.1f us  %5.1f%%" % (nm[:26], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
