"""The C++ host side: gm_graph API (clean room, reference-compatible), generated-style entry
points and benchmark drivers.  CPU tests need no GPU (they never touch the device mirror);
the driver runs are GPU tests."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLD, ROOT

PKG = os.path.join(ROOT, "green-marl_amd")
REF_APPS = "/root/reference/apps/output_cpp/src"
CXX_FLAGS = ["-O2", "-fopenmp", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "gm_graph", "inc"),
             "-I" + os.path.join(PKG, "generated"), "-I" + os.path.join(PKG, "apps")]
LINK = [os.path.join(PKG, "libgmgraph.a"), "-L" + PKG, "-lgmx", "-Wl,-rpath," + PKG, "-L/opt/rocm/lib",
        "-Wl,-rpath,/opt/rocm/lib", "-lamdhip64"]


@pytest.fixture(scope="module")
def host_built():
    subprocess.check_call(["make", "-C", PKG, "-j4", "lib", "host"], stdout=subprocess.DEVNULL)
    return PKG


def parse_dump(path):
    out = {}
    for line in open(path):
        k, *v = line.split()
        out[k] = np.array([int(x) for x in v], np.int64)
    return out


def test_gm_graph_api(host_built, golden, tmp_path):
    exe = str(tmp_path / "gm_graph_check")
    subprocess.check_call(["g++"] + CXX_FLAGS + [os.path.join(ROOT, "tests", "cpp", "gm_graph_check.cc"), "-o", exe] + LINK)
    src = os.path.join(GOLD, golden["manifest"]["bin"]["file"])
    out_bin, dump = str(tmp_path / "o.bin"), str(tmp_path / "d.txt")
    r = subprocess.run([exe, src, out_bin, dump], stdout=subprocess.PIPE, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout)
    assert "N = 256, M = 4096" in r.stdout                       # gm_graph_binary_loader.cc:136
    assert open(out_bin, "rb").read() == open(src, "rb").read()   # byte-identical with the reference's writer
    d = parse_dump(dump)
    c = golden["cases"]["rmat8_noperm"]
    for k in ("begin", "node_idx", "r_begin", "r_node_idx"):
        assert np.array_equal(d[k], c[k]), k
    # hand graph: rows sorted, reverse rows sorted by source
    assert d["h_begin"].tolist() == [0, 4, 4, 6, 7, 7, 8]
    assert d["h_node_idx"].tolist() == [1, 1, 3, 5, 0, 1, 3, 4]
    assert d["h_r_begin"].tolist() == [0, 1, 4, 4, 6, 7, 8]
    assert d["h_r_node_idx"].tolist() == [2, 0, 0, 2, 0, 3, 5, 0]
    assert d["h_node_idx_src"].tolist() == [0, 0, 0, 0, 2, 2, 3, 5]
    assert d["h_r_node_idx_src"].tolist() == [0, 1, 1, 1, 3, 3, 4, 5]
    # gm_rand32 sequence = the reference's update rule (gm_rand.cc:19-24) from its default seed
    x = np.int32(np.uint32(2463534242))
    want = []
    for _ in range(3):
        x = np.int32(np.uint32(x) ^ np.uint32((int(np.uint32(x)) << 13) & 0xffffffff))
        x = np.int32(x >> 17)
        x = np.int32(np.uint32(x) ^ np.uint32((int(np.uint32(x)) << 5) & 0xffffffff))
        want.append(int(x))
    assert d["rand32"].tolist() == want


@pytest.mark.skipif(not os.path.isdir(REF_APPS), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("app", ["pagerank", "hop_dist", "triangle_counting", "sssp", "avg_teen_cnt", "conduct", "bc"])
def test_reference_drivers_compile_unchanged(host_built, tmp_path, app):
    """Drop-in check: the REFERENCE's own driver sources (common_main.h + <app>_main.cc), untouched and
    compiled where they lie, build and link against this repo's gm.h / generated headers / libraries."""
    exe = str(tmp_path / app)
    flags = [f for f in CXX_FLAGS if "apps" not in f]   # the reference's common_main.h, not ours
    cmd = ["g++"] + flags + ["-I" + REF_APPS, "-w", os.path.join(REF_APPS, app + "_main.cc"), "-o", exe] + LINK
    subprocess.check_call(cmd)
    r = subprocess.run([exe], stdout=subprocess.PIPE, text=True)       # no args: usage line, exit(EXIT_FAILURE)
    assert r.returncode == 1 and "<graph_name> <num_threads> <nfspath>" in r.stdout


REF_GM = "/root/reference/apps/output_cpp/gm_graph"
REF_OBJ_DIR = os.path.join(ROOT, "oracle", "_ref")


def _build_option_a(tmp_path, app):
    """INTEGRATION.md Option A, literally: the reference's own compiled gm_graph objects (oracle/_ref/ref_*.o, built by
    oracle/Makefile from the sources where they lie; the test harness object is left out) + the reference's
    unchanged <app>_main.cc + the binding under green-marl_amd/integration/option_a compiled against the REFERENCE's
    headers + libgmx.  Only shl.h (Shoal's header, not vendored by the reference) comes from this repo."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-j4", "ref"], stdout=subprocess.DEVNULL)
    objs = sorted(os.path.join(REF_OBJ_DIR, f) for f in os.listdir(REF_OBJ_DIR)
                  if f.startswith("ref_") and f.endswith(".o") and f not in ("ref_harness.o", "ref_gm_default_usermain.o"))
    assert any(f.endswith("ref_gm_graph.o") for f in objs)
    inc = tmp_path / "inc"
    inc.mkdir(exist_ok=True)
    import shutil
    shutil.copy(os.path.join(PKG, "gm_graph", "inc", "shl.h"), inc / "shl.h")
    flags = ["-O2", "-fopenmp", "-std=gnu++11", "-w", "-I" + os.path.join(REF_GM, "inc"), "-I" + str(inc), "-I" + os.path.join(ROOT, "include"),
             "-I" + os.path.join(PKG, "generated"), "-I" + os.path.join(PKG, "integration", "option_a"), "-I" + REF_APPS]
    exe = str(tmp_path / ("option_a_" + app))
    cmd = (["g++"] + flags + [os.path.join(REF_APPS, app + "_main.cc"), os.path.join(PKG, "integration", "option_a", app + ".cc")] + objs +
           ["-o", exe, "-L" + PKG, "-lgmx", "-Wl,-rpath," + PKG, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lamdhip64"])
    subprocess.check_call(cmd)
    return exe


@pytest.mark.skipif(not os.path.isdir(REF_GM), reason="the reference checkout is not on this box")
@pytest.mark.parametrize("app", ["pagerank", "hop_dist", "triangle_counting"])
def test_option_a_links_against_the_reference_gm_graph(host_built, tmp_path, app):
    """Boundary proof for Option A: compile + link only; without a GPU the program still starts
    and prints the reference driver's usage line."""
    exe = _build_option_a(tmp_path, app)
    r = subprocess.run([exe], stdout=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "<graph_name> <num_threads> <nfspath>" in r.stdout


@pytest.mark.gpu
def test_option_a_programs_run_on_device(golden):
    """The Option-A programs built in the development container (oracle/Makefile `optiona`: the reference's drivers and
    gm_graph objects + the binding + libgmx) run on the GPU box: the reference's own load_binary reads the
    reference-written .bin fixture, the binding uploads gm_graph's arrays, and the printed results are the golden ones."""
    oa = os.path.join(REF_OBJ_DIR, "option_a")
    if not os.path.exists(os.path.join(oa, "pagerank")):
        pytest.skip("oracle/_ref/option_a was not built (needs the reference checkout at build time)")
    src = os.path.join(GOLD, golden["manifest"]["bin"]["file"])
    c = golden["cases"]["rmat8_noperm"]
    m = golden["manifest"]["rmat"]["rmat8_noperm"]

    def run(app):
        r = subprocess.run([os.path.join(oa, app), src, "4", "/dev/null"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r.returncode == 0 and "XXXXXXXXXX GM DONE XXXXXXXXXXXXXX" in r.stdout, r.stdout[-2000:]
        return r.stdout
    got = [float(x) for x in re.findall(r"rank\[\d\] = ([0-9.]+)", run("pagerank"))]
    assert got == [float("%0.9f" % x) for x in c["rank"][:4]]
    assert [int(x) for x in re.findall(r"dist\[\d\] = (\d+)", run("hop_dist"))] == c["dist"][:10].tolist()
    assert int(re.search(r"number of triangles: (\d+)", run("triangle_counting")).group(1)) == m["tc_directed"]


def run_app(app, *args):
    r = subprocess.run([os.path.join(PKG, "bin", app)] + list(args), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    assert "XXXXXXXXXX GM DONE XXXXXXXXXXXXXX" in r.stdout
    assert re.search(r"running time=\d+\.\d+", r.stdout)
    return r.stdout


@pytest.mark.gpu
def test_drivers_on_golden_bin(host_built, golden):
    src = os.path.join(GOLD, golden["manifest"]["bin"]["file"])
    c = golden["cases"]["rmat8_noperm"]
    m = golden["manifest"]["rmat"]["rmat8_noperm"]
    out = run_app("pagerank", src, "4", "/dev/null")
    got = [float(x) for x in re.findall(r"rank\[\d\] = ([0-9.]+)", out)]
    assert got == [float("%0.9f" % x) for x in c["rank"][:4]]
    out = run_app("pagerank", src, "4", "/dev/null", "100", "0.001", "0.85", "f32")
    got32 = [float(x) for x in re.findall(r"rank\[\d\] = ([0-9.]+)", out)]
    assert np.allclose(got32, c["rank"][:4], rtol=1e-5, atol=1e-9)
    out = run_app("hop_dist", src, "4", "/dev/null")
    assert [int(x) for x in re.findall(r"dist\[\d\] = (\d+)", out)] == c["dist"][:10].tolist()
    out = run_app("triangle_counting", src, "4", "/dev/null")
    assert int(re.search(r"number of triangles: (\d+)", out).group(1)) == m["tc_directed"]
    # sssp: the driver draws the edge lengths from gm_rand32 in slot order (sssp_main.cc:33-34); same stream here
    x = np.int32(np.uint32(2463534242).astype(np.int32))
    lens = []
    for _ in range(m["M"]):
        x = np.int32(np.uint32(x) ^ np.uint32((int(np.uint32(x)) << 13) & 0xffffffff))
        x = np.int32(x >> 17)
        x = np.int32(np.uint32(x) ^ np.uint32((int(np.uint32(x)) << 5) & 0xffffffff))
        lens.append(int(np.fmod(int(x), 100)) + 1)          # C's % truncates toward zero
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    g = po.Graph(m["N"], c["begin"].copy(), c["node_idx"].copy(), c["r_begin"].copy(), c["r_node_idx"].copy())
    want = po.sssp(g, np.array(lens, np.int32), 0)[0]
    out = run_app("sssp", src, "4", "/dev/null")
    assert [int(x) for x in re.findall(r"dist\[\d\] = (\d+)", out)] == want[:10].tolist()
    # avg_teen_cnt: every age is 10 (avg_teen_cnt_main.cc:16-18), K = 5 -> the average in-degree
    out = run_app("avg_teen_cnt", src, "4", "/dev/null")
    want_avg, _ = po.avg_teen_cnt(g, np.full(m["N"], 10, np.int32), 5)
    assert re.search(r"avg = ([0-9.]+)", out).group(1) == "%0.9f" % float(want_avg)
    # conduct: four groups from the gm_rand32 stream in vertex order (conduct_main.cc:27-38)
    x = np.int32(np.uint32(2463534242).astype(np.int32))
    member = []
    for _ in range(m["N"]):
        x = np.int32(np.uint32(x) ^ np.uint32((int(np.uint32(x)) << 13) & 0xffffffff))
        x = np.int32(x >> 17)
        x = np.int32(np.uint32(x) ^ np.uint32((int(np.uint32(x)) << 5) & 0xffffffff))
        r = int(np.fmod(int(x), 100))
        member.append(0 if r < 10 else 1 if r < 30 else 2 if r < 60 else 3)
    want_c = 0.0
    for i in range(4):
        want_c += float(po.conduct(g, np.array(member, np.int32), i))
    out = run_app("conduct", src, "4", "/dev/null")
    assert re.search(r"sum C = ([0-9.]+)", out).group(1) == "%f" % want_c


@pytest.mark.gpu
def test_driver_rmat_input(host_built, golden):
    m = golden["manifest"]["rmat"]["rmat14_noperm"]
    out = run_app("hop_dist", "RMAT:14:0", "4", "/dev/null")
    assert "N = 16384, M = 262144" in out
    out = run_app("pagerank", "RMAT:14:0", "4", "/dev/null")
    got = [float(x) for x in re.findall(r"rank\[\d\] = ([0-9.]+)", out)]
    assert got == [float("%0.9f" % x) for x in m["rank_head"]]


def test_uniform_generators_match_reference(host_built, golden, tmp_path):
    """create_uniform_random_graph_new / create_uniform_random_graph (graph_gen.cc:12-105: glibc rand() or the
    xorshift stream, rows filled from their last slot backwards) against the arrays the compiled reference wrote."""
    exe = str(tmp_path / "uniform_check")
    subprocess.check_call(["g++"] + CXX_FLAGS + [os.path.join(ROOT, "tests", "cpp", "uniform_check.cc"), "-o", exe] + LINK)
    n = 0
    for name, m in golden["manifest"]["uniform"].items():
        if "skipped" in m:
            continue
        dump = str(tmp_path / (name + ".txt"))
        subprocess.check_call([exe, str(m["N"]), str(m["M"]), str(m["seed"]), str(m["xorshift"]), dump])
        d = parse_dump(dump)
        c = golden["uniform"][name]
        assert np.array_equal(d["begin"], c["begin"]) and np.array_equal(d["node_idx"], c["node_idx"]), name
        n += 1
    assert n >= 2


def _unsorted_check(tmp_path, gpu):
    exe = str(tmp_path / "unsorted_check")
    subprocess.check_call(["g++"] + CXX_FLAGS + [os.path.join(ROOT, "tests", "cpp", "unsorted_check.cc"), "-o", exe] + LINK)
    r = subprocess.run([exe, str(tmp_path / "unsorted.bin")] + (["gpu"] if gpu else []), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout)
    return r.stdout


def test_load_binary_semi_sorts_unsorted_rows(host_built, tmp_path):
    """load_binary of a file with unsorted rows: sorted rows + e_idx2idx + reverse CSR (host path on this box)."""
    assert "load_binary of unsorted rows ok" in _unsorted_check(tmp_path, False)


@pytest.mark.gpu
def test_unsorted_host_graph_on_device(host_built, tmp_path):
    """The same through the device row sort, and sssp through the generated entry on a host graph that is frozen
    but not semi-sorted: the caller's len[] has to follow the device's row sort."""
    out = _unsorted_check(tmp_path, True)
    assert "load_binary of unsorted rows ok" in out and "sssp on an unsorted host graph ok" in out


@pytest.mark.gpu
def test_driver_pagerank_same_lines_for_any_rank_count(host_built):
    """bin/pagerank (the emitted driver, unchanged) on one rank and on several rank states driven from the one
    host thread prints the same rank[i] lines (9 decimals) and the same iteration behaviour."""
    def lines(env):
        r = subprocess.run([os.path.join(PKG, "bin", "pagerank"), "RMAT:20:1", "16", "/dev/null"], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, env=dict(os.environ, **env), timeout=600)
        assert r.returncode == 0, r.stdout[-2000:]
        return re.findall(r"rank\[\d\] = [0-9.]+", r.stdout)
    one = lines({"GMX_DEVICES": "1", "GMX_PR_RANKS": "1"})
    assert len(one) == 4
    assert lines({"GMX_PR_RANKS": "4"}) == one
    assert lines({}) == one                      # every visible device, one rank each


@pytest.mark.gpu
def test_gm_graph_api_with_device(host_built, golden, tmp_path):
    """Same API check on a box with a GPU: load_binary then builds the reverse CSR on the device."""
    test_gm_graph_api(host_built, golden, tmp_path)


@pytest.mark.gpu
def test_gmx_bench_tool(host_built):
    r = subprocess.run([os.path.join(PKG, "bin", "gmx_bench")], stdout=subprocess.PIPE, text=True)
    assert r.returncode == 0 and "gfx950" in r.stdout


# ---------------------------------------------------------------- GM_EDGE64 host build (edge_t = int64_t)
LINK64 = [os.path.join(PKG, "libgmgraph_e64.a")] + LINK[1:]


@pytest.fixture(scope="module")
def host64_built(host_built):
    subprocess.check_call(["make", "-C", PKG, "-j4", "host64"], stdout=subprocess.DEVNULL)
    return PKG


def _e64_file(host64_built, golden, tmp_path):
    """gm_graph_check built with -DGM_EDGE64: loads the committed 32-bit file (a wider library reads the narrower file,
    gm_graph_binary_loader.cc:93-101) and stores it with 8-byte edge fields."""
    exe = str(tmp_path / "gm_graph_check64")
    subprocess.check_call(["g++"] + CXX_FLAGS + ["-DGM_EDGE64", os.path.join(ROOT, "tests", "cpp", "gm_graph_check.cc"), "-o", exe] + LINK64)
    src = os.path.join(GOLD, golden["manifest"]["bin"]["file"])
    out_bin, dump = str(tmp_path / "o64.bin"), str(tmp_path / "d64.txt")
    r = subprocess.run([exe, src, out_bin, dump], stdout=subprocess.PIPE, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout)
    return exe, out_bin, parse_dump(dump)


def test_edge64_host_build(host64_built, golden, tmp_path):
    """The host side built for edge_t = int64_t (the reference's EDGE_SIZE=64, common.mk:46-52): same API results as the
    32-bit build, the .bin format with 8-byte edge fields (gm_graph_binary_loader.cc:19-26, :222-252), and the link-time
    size check.  The reference itself does not compile in this configuration (gm_graph.cc:226), so the 64-bit bytes are
    checked against the documented layout, not against a reference-written file: parity unpinned for that file."""
    exe, out_bin, d = _e64_file(host64_built, golden, tmp_path)
    c = golden["cases"]["rmat8_noperm"]
    for k in ("begin", "node_idx", "r_begin", "r_node_idx"):
        assert np.array_equal(d[k], c[k]), k
    raw = open(out_bin, "rb").read()
    N, M = 256, 4096
    assert len(raw) == 12 + 4 + 8 + 8 * (N + 1) + 4 * M
    assert np.frombuffer(raw, ">u4", 4).tolist() == [0x03939999, 4, 8, N]
    assert np.frombuffer(raw, ">i8", 1, 16)[0] == M
    assert np.array_equal(np.frombuffer(raw, ">i8", N + 1, 24), c["begin"])
    assert np.array_equal(np.frombuffer(raw, ">i4", M, 24 + 8 * (N + 1)), c["node_idx"])
    # the 64-bit library reads its own file back (same arrays, same bytes out) ...
    out2, dump2 = str(tmp_path / "o64b.bin"), str(tmp_path / "d64b.txt")
    r = subprocess.run([exe, out_bin, out2, dump2], stdout=subprocess.PIPE, text=True)
    assert r.returncode == 0 and "N = 256, M = 4096" in r.stdout, (r.returncode, r.stdout)
    assert open(out2, "rb").read() == raw
    assert np.array_equal(parse_dump(dump2)["r_node_idx"], c["r_node_idx"])
    # ... and the 32-bit library refuses it, like the reference (:97-100)
    exe32 = str(tmp_path / "gm_graph_check32")
    subprocess.check_call(["g++"] + CXX_FLAGS + [os.path.join(ROOT, "tests", "cpp", "gm_graph_check.cc"), "-o", exe32] + LINK)
    r = subprocess.run([exe32, out_bin, str(tmp_path / "x.bin"), str(tmp_path / "x.txt")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 1 and "size mismatch" in r.stdout
    # an application built for 32-bit edges does not link against the 64-bit library (gm_graph_typedef.h size check)
    r = subprocess.run(["g++"] + CXX_FLAGS + [os.path.join(ROOT, "tests", "cpp", "gm_graph_check.cc"), "-o", str(tmp_path / "bad")] + LINK64,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode != 0 and "node32_edge32" in r.stdout


@pytest.mark.gpu
def test_edge64_drivers_on_device(host64_built, golden, tmp_path):
    """bin64/* (the drivers and generated entries compiled with -DGM_EDGE64) on the 64-bit file: the device mirror goes
    through gmx_graph_upload_e64, load_binary's device preparation through the other *_e64 calls; same output lines as
    the 32-bit build on the 32-bit file."""
    _, out_bin, d = _e64_file(host64_built, golden, tmp_path)
    c = golden["cases"]["rmat8_noperm"]
    assert np.array_equal(d["r_begin"], c["r_begin"]) and np.array_equal(d["r_node_idx"], c["r_node_idx"])
    src32 = os.path.join(GOLD, golden["manifest"]["bin"]["file"])
    keep = re.compile(r"^(rank\[|dist\[|number of triangles|N = )")
    for app in ("pagerank", "hop_dist", "triangle_counting"):
        outs = []
        for exe, f in ((os.path.join(PKG, "bin64", app), out_bin), (os.path.join(PKG, "bin", app), src32)):
            r = subprocess.run([exe, f, "4", "/dev/null"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
            assert r.returncode == 0, r.stdout[-2000:]
            outs.append([l for l in r.stdout.splitlines() if keep.match(l)])
        assert outs[0] == outs[1] and len(outs[0]) >= 2, (app, outs)


@pytest.mark.gpu
def test_edge64_capi_limits_and_round_trip(golden):
    """gmx_graph_upload_e64 / download_e64 / edge_order_e64 / reverse_edge_map_e64 against the 32-bit calls; refusals."""
    import ctypes as C
    sys.path.insert(0, PKG)
    import gmx
    L = gmx.lib()
    c = golden["cases"]["rmat8_noperm"]
    N, M = 256, 4096
    b64 = np.ascontiguousarray(c["begin"], np.int64)
    idx = np.ascontiguousarray(c["node_idx"], np.int32)
    h = C.c_void_p()
    assert L.gmx_graph_upload_e64(b64.ctypes.data, idx.ctypes.data, None, None, N, M, 0, C.byref(h)) == 0
    ob, orb = np.zeros(N + 1, np.int64), np.zeros(N + 1, np.int64)
    oi, ori = np.zeros(M, np.int32), np.zeros(M, np.int32)
    assert L.gmx_graph_download_e64(h, ob.ctypes.data, oi.ctypes.data, orb.ctypes.data, ori.ctypes.data) == 0
    assert np.array_equal(ob, c["begin"]) and np.array_equal(oi, c["node_idx"])
    assert np.array_equal(orb, c["r_begin"]) and np.array_equal(ori, c["r_node_idx"])
    m64, m32 = np.zeros(M, np.int64), np.zeros(M, np.int32)
    assert L.gmx_graph_reverse_edge_map_e64(h, m64.ctypes.data) == 0 and L.gmx_graph_reverse_edge_map(h, m32.ctypes.data) == 0
    assert np.array_equal(m64, m32)
    ident = C.c_int(0)
    assert L.gmx_graph_edge_order_e64(h, m64.ctypes.data, C.byref(ident)) == 0 and ident.value == 1
    L.gmx_graph_free(h)
    # refusals: an edge count the 32-bit device offsets cannot hold; offsets outside [0, E]
    h2 = C.c_void_p()
    assert L.gmx_graph_upload_e64(b64.ctypes.data, idx.ctypes.data, None, None, N, 1 << 31, 0, C.byref(h2)) != 0 and not h2.value
    assert b"32-bit edge offsets" in L.gmx_last_error()
    bad = b64.copy()
    bad[5] = 1 << 40
    assert L.gmx_graph_upload_e64(bad.ctypes.data, idx.ctypes.data, None, None, N, M, 0, C.byref(h2)) != 0 and not h2.value
