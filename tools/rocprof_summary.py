"""Summarise a rocprofv3 rocpd sqlite database (``*_results.db``) per kernel:
calls, total / average / min / max duration.  Usage: rocprof_summary.py DB [DB...]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r'\(.*', '', name)
    return name if len(name) <= 70 else name[:67] + '...'


def main(paths):
    for path in paths:
        c = sqlite3.connect(path)
        rows = c.execute('select name, count(*), sum(duration), avg(duration), min(duration), '
                         'max(duration) from kernels group by name order by sum(duration) desc').fetchall()
        total = sum(r[2] for r in rows) or 1
        print(f'# {path}: {sum(r[1] for r in rows)} dispatches, {total / 1e6:.3f} ms of kernel time')
        print(f'{"kernel":70s} {"calls":>6s} {"total_ms":>10s} {"avg_us":>10s} {"min_us":>9s} '
              f'{"max_us":>9s} {"pct":>6s}')
        for name, n, tot, avg, mn, mx in rows:
            print(f'{short(name):70s} {n:6d} {tot / 1e6:10.3f} {avg / 1e3:10.2f} {mn / 1e3:9.2f} '
                  f'{mx / 1e3:9.2f} {100 * tot / total:6.2f}')


if __name__ == '__main__':
    main(sys.argv[1:])
