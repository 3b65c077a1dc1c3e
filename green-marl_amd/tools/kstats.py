#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace result database (rocpd sqlite): count, total, average.
usage: kstats.py <results.db> [name-substring ...]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    pats = sys.argv[2:]
    where = " or ".join("name like '%%%s%%'" % p for p in pats) if pats else "1"
    q = "select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 from kernels where %s group by name order by 3 desc" % where
    print("%-72s %6s %12s %10s %10s %10s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us"))
    for name, n, tot, avg, mn, mx in db.execute(q):
        print("%-72s %6d %12.1f %10.1f %10.1f %10.1f" % (name[:72], n, tot, avg, mn, mx))


if __name__ == "__main__":
    main()
