# per-launch kernel times of hop_dist RMAT-26 from vertex 0 (development helper; run through gpurun)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/bfs26 -- python3 green-marl_amd/tools/bfs_prof.py 26 3 > gpurun_out/bfs26.log 2>&1 || exit 1
tail -3 gpurun_out/bfs26.log
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/bfs26/*/*kernel_trace.csv"))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if r["Kernel_Name"].startswith("bfs") or "bfs_" in r["Kernel_Name"]]
# the last traversal: from the last bfs_init_kernel on
last = max(i for i, r in enumerate(rows) if "bfs_init_kernel" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    print("%9.1f us  +%8.1f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"].split("(")[0]))
PY
