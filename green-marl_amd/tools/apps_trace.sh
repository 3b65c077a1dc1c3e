cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/apps_trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/apps_trace -- python3 green-marl_amd/tools/apps_prof.py 24 > gpurun_out/apps_trace.log 2>&1
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/apps_trace/*/*kernel_trace.csv"))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "pred_bitmap_kernel" in r["Kernel_Name"]]
for start in (idx[1], idx[-1]):     # second avg_teen call, last conduct call
    t0 = int(rows[start - 3]["Start_Timestamp"]); prev = t0
    for r in rows[start - 3:start + 5]:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%9.1f us +%8.1f gap %7.1f %s" % ((st - t0) / 1e3, (en - st) / 1e3, (st - prev) / 1e3, r["Kernel_Name"].split("(")[0][:60]))
        prev = en
    print("--")
PY
