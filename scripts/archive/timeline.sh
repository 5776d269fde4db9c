# start/end of every libkmx kernel of the last bench steps (rocprofv3 kernel trace): where the step time goes between merges
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/timeline
rm -rf $OUT; mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 6 > $OUT/log.txt 2>&1
python - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/timeline/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "kmx::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-30:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    n = r["Kernel_Name"].split("kmx::")[1].split("(")[0][:22]
    print(f"{n:24s} q={r.get('Queue_Id','?'):>3s} start {(int(r['Start_Timestamp'])-t0)/1e6:8.3f} ms  end {(int(r['End_Timestamp'])-t0)/1e6:8.3f} ms  dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6:6.3f}")
PY
find $OUT -name "*.csv" -size +5M -delete
python - <<'PY'
import csv, glob, os
fs = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/timeline/**/*memory_copy_trace.csv", recursive=True)
kf = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/timeline/**/*kernel_trace.csv", recursive=True)[0]
ks = [r for r in csv.DictReader(open(kf)) if "k_merge_cols" in r["Kernel_Name"]]
ks.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(ks[-3]["Start_Timestamp"])
if fs:
    rows = list(csv.DictReader(open(fs[0])))
    print(rows[0].keys())
    for r in rows:
        s = (int(r["Start_Timestamp"]) - t0) / 1e6
        if s > -0.5: print("copy", r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?")), f"start {s:8.3f} end {(int(r['End_Timestamp'])-t0)/1e6:8.3f}")
print("cols kernels at", [(int(k["Start_Timestamp"]) - t0) / 1e6 for k in ks[-3:]])
PY
