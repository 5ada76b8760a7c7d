#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace run (rocpd .db or kernel_trace csv)
into a per-kernel table (calls, total/avg/min/max ms, share)."""
import glob
import os
import sqlite3
import sys


def main(path, out=None):
    dbs = glob.glob(os.path.join(path, '**', '*.db'), recursive=True)
    rows = []
    for db in dbs:
        con = sqlite3.connect(db)
        rows += con.execute(
            'select name, count(*), sum(end-start)/1e6, avg(end-start)/1e6, '
            'min(end-start)/1e6, max(end-start)/1e6 from kernels group by name').fetchall()
    rows.sort(key=lambda r: -r[2])
    tot = sum(r[2] for r in rows) or 1.0
    lines = ['%-100s %6s %11s %10s %10s %10s %6s' % ('kernel', 'calls', 'total_ms', 'avg_ms',
                                                      'min_ms', 'max_ms', '%')]
    for r in rows:
        lines.append('%-100s %6d %11.3f %10.4f %10.4f %10.4f %6.2f' % (
            r[0][:100], r[1], r[2], r[3], r[4], r[5], 100*r[2]/tot))
    text = '\n'.join(lines)
    print(text)
    if out:
        with open(out, 'w') as f:
            f.write(text + '\n')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
