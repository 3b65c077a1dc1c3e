#!/usr/bin/env python3
"""exchange_stats.py -- how much of a rank's contribution range do the OTHER ranks of an N-rank PageRank partition read?

Rows are dealt to ranks by position in the descending-out-degree order (position j -> rank j % N, gmx_pagerank.hip);
rank q reads source w iff w has an out-edge into a row q owns.  Prints, for N in 2, 4, 8: live sources, the
(source, reader) incidences excluding the owner, and the bytes per step and rank that a "send only what is read"
exchange moves against the full-prefix exchange.  Run on a GPU box:  python exchange_stats.py [scale]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gmx  # noqa: E402


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
    gmx.require_device()
    N, M = 1 << scale, 16 << scale
    g = gmx.Graph.rmat(N, M, 1997, 0.57, 0.19, 0.19, True)
    begin, node_idx, _, _ = g.download()
    g.free()
    deg = np.diff(begin).astype(np.int64)
    order = np.argsort(-deg, kind="stable")
    pos = np.empty(N, np.int64)
    pos[order] = np.arange(N)
    live = int((deg > 0).sum())
    print("RMAT-%d: V %d, E %d, live sources %d (%.1f %%)" % (scale, N, M, live, 100.0 * live / N))
    hist = np.bincount(np.minimum(deg, 64))
    print("out-degree histogram (capped at 64):", hist[:17].tolist(), "...", int(hist[17:].sum()))
    starts = begin[:-1].astype(np.int64)
    nz = deg > 0
    for nr in (2, 4, 8):
        owner_of_row = (pos % nr).astype(np.int8)
        dst_owner = owner_of_row[node_idx]
        readers = np.zeros(N, np.int64)          # number of distinct ranks that read the source
        foreign = np.zeros(N, np.int64)          # ... other than its owner
        for r in range(nr):
            hit = (dst_owner == r).astype(np.int8)
            cnt = np.zeros(N, np.int64)
            cnt[nz] = np.add.reduceat(hit, starts[nz])
            rd = cnt > 0
            readers += rd
            foreign += rd & (owner_of_row != r)
        inc = int(foreign.sum())
        full = live * (nr - 1)
        print("N = %d: (source, foreign reader) incidences %d of %d (%.1f %%): %.1f MB in per rank and step packed, %.1f MB full prefix; "
              "sources read by every rank %.1f %%, by one rank only %.1f %%"
              % (nr, inc, full, 100.0 * inc / full, inc * 4 / nr / 1e6, full * 4 / nr / 1e6,
                 100.0 * float((readers[nz] == nr).sum()) / live, 100.0 * float((readers[nz] == 1).sum()) / live))


if __name__ == "__main__":
    main()
