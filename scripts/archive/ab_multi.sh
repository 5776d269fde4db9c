# A/B/... of several prebuilt libkmx builds on one box: kmtricks_amd/libkmx_<name>.so.bin for every name in $VARIANTS, command in "$@"
cd $GRAFT_REPO_ROOT
for round in 1 2 3; do for v in $VARIANTS; do cp kmtricks_amd/libkmx_$v.so.bin kmtricks_amd/libkmx.so; echo -n "$v: "; "$@"; done; done
