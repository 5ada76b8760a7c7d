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
    # how much of the trace's span the GPU had a kernel running (union of the kernels' intervals)
    for db in dbs:
        iv = sqlite3.connect(db).execute('select start, end from kernels order by start').fetchall()
        if not iv:
            continue
        busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
        for a, b in iv[1:]:
            if a > cur_e:
                busy += cur_e - cur_s
                cur_s, cur_e = a, b
            else:
                cur_e = max(cur_e, b)
        busy += cur_e - cur_s
        span = max(b for _, b in iv) - iv[0][0]
        lines.append('# %s: kernels running %.1f ms of the %.1f ms between the first kernel\'s start '
                     'and the last one\'s end (%.0f %%); summed kernel durations %.1f ms' % (
                         os.path.basename(db), busy/1e6, span/1e6, 100.0*busy/span, tot))
    if os.environ.get('ROCPROF_GAPS'):
        # where the GPU waits: idle time between consecutive kernels, by (kernel before, after)
        import collections
        for db in dbs:
            iv = sqlite3.connect(db).execute(
                'select start, end, name from kernels order by start').fetchall()
            gaps = collections.defaultdict(lambda: [0, 0.0])
            cur_e, cur_n = iv[0][1], iv[0][2]
            for a, b, nm in iv[1:]:
                if a > cur_e:
                    if a - cur_e < 50e6:   # (not the pauses between the run's phases)
                        g = gaps[(cur_n[:48], nm[:48])]
                        g[0] += 1
                        g[1] += (a - cur_e)/1e6
                if b > cur_e:
                    cur_e, cur_n = b, nm
            lines.append('# idle between kernels (count, total ms): before -> after')
            for (k0, k1), (cnt, ms) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:16]:
                lines.append('#   %5d %9.2f  %s -> %s' % (cnt, ms, k0, k1))
    text = '\n'.join(lines)
    print(text)
    if out:
        with open(out, 'w') as f:
            f.write(text + '\n')




def pmc(path):
    """Per-kernel mean of every collected PMC counter (rocprofv3 --pmc run)."""
    import collections
    dbs = glob.glob(os.path.join(path, '**', '*.db'), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for db in dbs:
        con = sqlite3.connect(db)
        try:
            cols = [r[1] for r in con.execute('pragma table_info(counters_collection)')]
            rows = con.execute('select * from counters_collection').fetchall()
        except sqlite3.Error as e:
            print('no counters_collection view:', e)
            continue
        ci = {c: i for i, c in enumerate(cols)}
        namec = 'kernel_name' if 'kernel_name' in ci else 'name'
        for r in rows:
            acc[r[ci[namec]]][r[ci['counter_name']]].append(r[ci['value']])
    for k, d in sorted(acc.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
        print(k[:110])
        for cname, vals in d.items():
            print('    %-20s calls %4d  mean %.6g  total %.6g  min %.6g  max %.6g' % (
                cname, len(vals), sum(vals)/len(vals), sum(vals), min(vals), max(vals)))


if __name__ == '__main__':
    if sys.argv[1] == '--pmc':
        pmc(sys.argv[2])
    else:
        main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
