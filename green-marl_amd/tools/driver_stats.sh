# rocprofv3 kernel stats of one run of the emitted driver bin/pagerank on RMAT-24 (where does "running time" go)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT/green-marl_amd
rocprofv3 --kernel-trace --stats --output-format csv -d ../gpurun_out/drv -- ./bin/pagerank RMAT:24:1 16 /dev/null > ../gpurun_out/drv.log 2>&1
grep -i "time" ../gpurun_out/drv.log
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("../gpurun_out/drv/*/*kernel_stats.csv"))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms" % (tot / 1e6))
for r in rows[:22]:
    n = r["Name"]
    n = n[:n.index("(")] if "(" in n else n
    if "rocprim" in n:
        i = n.find("detail::", n.find("trampoline_kernel"))
        n = "rocprim " + ("radix_sort" if "radix_sort" in r["Name"] else "partition" if "partition" in r["Name"] else "scan" if "scan" in r["Name"] else "other")
    print("%-50s calls %4s total %8.2f ms" % (n[:50], r["Calls"], float(r["TotalDurationNs"]) / 1e6))
PY
