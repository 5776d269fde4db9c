# A/B of two prebuilt libkmx builds on one box: kmtricks_amd/libkmx_{base,new}.so.bin, command in "$@"
cd $GRAFT_REPO_ROOT
for round in 1 2 3; do for v in base new; do cp kmtricks_amd/libkmx_$v.so.bin kmtricks_amd/libkmx.so; echo -n "$v: "; "$@"; done; done
