#!/usr/bin/env python3
"""A/B timing of the PageRank step of two builds of libgmx.so on the same box (alternating, same graph size).
usage: ab_pr.py <libA.so> <libB.so> [scale] [nranks] [rounds]"""
import ctypes as C
import sys
import time


def load(path):
    L = C.CDLL(path)
    vp, i64 = C.c_void_p, C.c_int64
    L.gmx_graph_create_rmat.argtypes = [i64, i64, C.c_long, C.c_double, C.c_double, C.c_double, C.c_int, C.c_uint32, C.POINTER(vp)]
    L.gmx_pr_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_uint32, C.POINTER(vp)]
    L.gmx_pr_reset.argtypes = [vp, C.c_double]
    L.gmx_pr_step.argtypes = [vp, vp]
    L.gmx_pr_diff.argtypes = [vp, vp, C.POINTER(C.c_double)]
    L.gmx_pr_free.argtypes = [vp]
    L.gmx_graph_free.argtypes = [vp]
    L.gmx_last_error.restype = C.c_char_p
    return L


ELEM = int(__import__("os").environ.get("AB_ELEM", "4"))


def main():
    pa, pb = sys.argv[1], sys.argv[2]
    scale = int(sys.argv[3]) if len(sys.argv) > 3 else 26
    nranks = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 3
    libs = [("A", load(pa)), ("B", load(pb))]
    state = []
    for name, L in libs:
        g, p = C.c_void_p(), C.c_void_p()
        assert L.gmx_graph_create_rmat(1 << scale, 16 << scale, 1997, 0.57, 0.19, 0.19, 1, 0, C.byref(g)) == 0, L.gmx_last_error()
        assert L.gmx_pr_create(g, ELEM, 0, nranks, 7, C.byref(p)) == 0, L.gmx_last_error()
        L.gmx_pr_reset(p, 0.85)
        state.append((name, L, g, p))
    d = C.c_double(0)
    for r in range(rounds):
        for name, L, g, p in state:
            for _ in range(3):
                L.gmx_pr_step(p, None)
            L.gmx_pr_diff(p, None, C.byref(d))
            t0 = time.perf_counter()
            for _ in range(20):
                L.gmx_pr_step(p, None)
            L.gmx_pr_diff(p, None, C.byref(d))
            ms = (time.perf_counter() - t0) * 1e3 / 20
            print("round %d lib %s: %.3f ms/iter (diff %.3e)" % (r, name, ms, d.value), flush=True)
    for name, L, g, p in state:
        L.gmx_pr_free(p)
        L.gmx_graph_free(g)


if __name__ == "__main__":
    main()
