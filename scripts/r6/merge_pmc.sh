#!/bin/bash
# round 6: SQ counters of a merge workload's kernels (WL = count_200 | count | pa63 | bft | bf): how busy the vector units are
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6mpmc; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for P in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  timeout 900 rocprofv3 --pmc $P -d $O/p --output-format csv -- python $R/bench.py --workload ${WL:-count_200} --steps 3 --warmup 1 --no-cpu-baseline --no-whole-job > $O/log.txt 2>&1
  python - <<PY
import glob, csv, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$O/p/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_merge" in n or "k_cols" in n: per[(r["Dispatch_Id"], n.split("(")[0][-40:], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, k, c), v in per.items():
        a = acc[(k, c)]; a[0] += v; a[1] += 1
for (k, c), (s, n) in sorted(acc.items()): print(k, c, round(s / n), n)
PY
  rm -rf $O/p
done
