# rocprofv3 counter passes over the merge bench (one counter group per pass, no trace domains);
# prints the per-launch mean of every counter for kernels whose name contains $1 (default k_merge)
KN=${1:-k_merge}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
i=0
for G in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
         "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $G -d $OUT/p$i --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/p$i.log 2>&1
done
python - "$KN" <<'PY'
import sys, glob, csv, collections
kn = sys.argv[1]
acc = collections.defaultdict(list)
import os
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc/p*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if kn in r["Kernel_Name"]:
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, c), v in per.items():
        acc[c].append(v)
for c in sorted(acc):
    v = acc[c]
    print(f"{c:24s} mean={sum(v)/len(v):.6g}  n={len(v)}")
PY
find $OUT -name "*.csv" -size +20M -delete
