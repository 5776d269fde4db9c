# phase profile (clock64 per phase, -DKMX_PHASE_PROF build made by scripts/dev/build_variant.sh prof) of k_merge_cols / k_cols_sparse
# on the count and pa63 workloads, rows in file order (1) or where the kernels leave them (0)
for wl in ${WLS:-count pa63}; do for fo in ${FOS:-1 0}; do
  echo "== $wl file_order=$fo"
  KMX_LIB=$GRAFT_REPO_ROOT/kmtricks_amd/libkmx_prof.so KMX_FILE_ORDER=$fo python bench.py --workload $wl --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "^\[cols\]|^\[sparse\]" | tail -14
done; done
