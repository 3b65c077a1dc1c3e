#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in path: the host hands over gm_graph's CSR arrays (as the generated
pagerank(gm_graph&, ...) does through gmx_graph_upload), the ranks come back to the host.
usage: pcie_rate.py [scale] [iterations]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gmx


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    g = gmx.Graph.rmat(1 << scale, 16 << scale, 1997, 0.57, 0.19, 0.19, True)
    begin, node_idx, rb, rn = g.download()
    E = g.E
    g.free()
    for rep in range(2):
        t0 = time.perf_counter()
        h = gmx.Graph.upload(begin, node_idx, rb, rn)
        t1 = time.perf_counter()
        rank, st = h.pagerank(1e-300, 0.85, iters, np.float32)
        t2 = time.perf_counter()
        h.free()
    up, run = t1 - t0, t2 - t1
    gb = (begin.nbytes + node_idx.nbytes + rb.nbytes + rn.nbytes) / 1e9
    print("RMAT-%d: upload of %.2f GB host CSR %.1f ms (%.1f GB/s); pagerank fp32 %d iterations incl. plan and rank download %.1f ms "
          "(device loop %.1f ms, d2h %.2f ms)" % (scale, gb, up * 1e3, gb / up, st["iterations"], run * 1e3, st["kernel_ms"], st["d2h_ms"]))
    print("  PCIe-inclusive: %.1f GTEPS over upload + call (%.1f GTEPS over the call alone; %.1f GTEPS device loop)"
          % (E * st["iterations"] / (up + run) / 1e9, E * st["iterations"] / run / 1e9, E * st["iterations"] / (st["kernel_ms"] * 1e-3) / 1e9))


if __name__ == "__main__":
    main()
