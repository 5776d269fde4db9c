#!/bin/bash
# round 5: what bounds the column-blocked pair in file order -- SQ counters per kernel (rocprofv3 --pmc, passes of their own, no trace
# domains) over a few batches of configs[2]'s lists (scripts/r5_grid.py with one setting), then the clock64 phase profile of the same
# (libkmx_prof.so: scripts/dev/build_variant.sh prof "-DKMX_PHASE_PROF").   WL=count|pa63
set -x
cd /tmp && export TMPDIR=/tmp
WL=${WL:-count}
O=$GRAFT_REPO_ROOT/gpurun_out/r5b_$WL
rm -rf $O; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|TCP_[A-Z_0-9]*\|TCC_[A-Z_0-9]*\|GRBM_[A-Z_0-9]*" | sort -u > $O/counters.txt
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH"
P2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
P3="SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"
P4="GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_I8 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_VALU SQ_INSTS_SENDMSG SQ_WAVES_EQ_64 SQ_IFETCH"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $P -d $O/p$i --output-format csv -- python $GRAFT_REPO_ROOT/scripts/r5_grid.py --wl $WL --steps 3 --settings "x=1" > $O/p$i.log 2>&1
done
python - <<'PY' | tee $O/sq_counters.txt
import glob, csv, collections, os
O = os.environ.get("O") or glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r5b_*")[0]
acc = collections.defaultdict(list)
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r5b_" + os.environ.get("WL", "count") + "/p*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "k_merge_cols" in kn or "k_cols_sparse" in kn:
            short = kn.split("(")[0].split("::")[-1]
            per[(r["Dispatch_Id"], short, r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, k, c), v in per.items():
        acc[(k, c)].append(v)
for k in sorted(acc):
    v = acc[k]
    print(f"{k[0]:44s} {k[1]:24s} mean={sum(v)/len(v):.6g}  n={len(v)}")
PY
find $O -name "*.csv" -size +5M -delete
cd $GRAFT_REPO_ROOT && KMX_LIB=$GRAFT_REPO_ROOT/kmtricks_amd/libkmx_prof.so python scripts/r5_grid.py --wl $WL --steps 4 --settings "x=1" > $O/phase.jsonl 2> $O/phase.err
grep -E "^\[cols\]|^\[sparse\]" $O/phase.err | tail -56
