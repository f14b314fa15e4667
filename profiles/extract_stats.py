"""rocprofv3 (ROCm 7.2) writes a rocpd sqlite database by default; dump its
`top_kernels` view (what `--stats` prints) as CSV so that it can be committed.
usage: python profiles/extract_stats.py <results.db> <out.csv>"""
import csv
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
rows = list(cur.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
with open(sys.argv[2], 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['Name', 'Calls', 'TotalDurationUs', 'AverageUs', 'Percentage'])
    w.writerows(rows)
print(f'{len(rows)} kernels -> {sys.argv[2]}')
