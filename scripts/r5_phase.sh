#!/bin/bash
# clock64 phase profile of the column-blocked pair on configs[2]'s lists, file order first (scripts/r5_grid.py prints the steps; the
# dumps come on stderr at every result's free): LIBS="prof prof0" = libkmx_<name>.so built by scripts/dev/build_variant.sh
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5p; mkdir -p $O
for v in ${LIBS:-prof}; do
  KMX_LIB=$GRAFT_REPO_ROOT/kmtricks_amd/libkmx_$v.so python scripts/r5_grid.py --wl ${WL:-count} --steps 3 --settings "x=1" > $O/$v.jsonl 2> $O/$v.err
  echo "== $v (first dumps: file order)"; grep -E "^\[cols\]|^\[sparse\]" $O/$v.err | sed -n '29,42p'
done
