#!/bin/bash
# round 5: variant builds of libkmx (scripts/dev/build_variant.sh <name> "<flags>") side by side on one box: the pair's two kernels by
# HIP events (scripts/r5_grid.py, default setting only).  VARIANTS="name name ..." ("base" = the product's libkmx.so); WL=count|pa63
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5v; mkdir -p $O
for v in ${VARIANTS:-base}; do
  lib=$GRAFT_REPO_ROOT/kmtricks_amd/libkmx_$v.so; [ "$v" = base ] && lib=$GRAFT_REPO_ROOT/kmtricks_amd/libkmx.so
  for r in 1 ${REPEAT:+2}; do
    KMX_LIB=$lib python scripts/r5_grid.py --wl ${WL:-count} --steps ${STEPS:-8} --settings "x=1" 2> $O/$v.err | sed "s/^{/{\"variant\": \"$v\", /" | tee -a $O/variants_${WL:-count}.jsonl
  done
done
