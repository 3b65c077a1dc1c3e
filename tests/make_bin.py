"""Test helper (not product): write an RMAT graph generated on the device as a reference-format .bin file
with the oracle's writer.  usage: make_bin.py <scale> <permute 0|1> <out.bin>"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(HERE, "..", "green-marl_amd"), os.path.join(HERE, "..", "oracle")):
    sys.path.insert(0, os.path.abspath(p))
import gmx
import pyoracle as po

scale, permute, out = int(sys.argv[1]), bool(int(sys.argv[2])), sys.argv[3]
g = gmx.Graph.rmat(1 << scale, 16 << scale, 1997, 0.57, 0.19, 0.19, permute)
begin, node_idx, rb, rn = g.download()
g.free()
po.store_binary(out, po.Graph(1 << scale, begin, node_idx, rb, rn))
print("wrote", out, os.path.getsize(out), "bytes")
