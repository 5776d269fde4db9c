#!/bin/bash
# round 6: SQ counters of one count-stage kernel (KERNEL, a substring of its name) under two environments (SETS as count_env_ab.sh)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6cpmc; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
IFS=';' read -ra SETS_A <<< ";${SETS}"
i=0; for S in "${SETS_A[@]}"; do i=$((i+1))
  for P in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
    env $S timeout 600 rocprofv3 --pmc $P -d $O/p --output-format csv -- python $R/scripts/bench_count_stage.py --genome 5e6 --partitions 256 --reps 4 --skip-streams > $O/log.txt 2>&1
    python - <<PY
import glob, csv, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$O/p/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "${KERNEL:-k_cs_wave_sort}" in r["Kernel_Name"]: per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, c), v in per.items():
        a = acc[c]; a[0] += v; a[1] += 1
print("[${S:-defaults}]", {c: round(s / n) for c, (s, n) in sorted(acc.items())})
PY
    rm -rf $O/p
  done
done
