set -e
cd $GRAFT_REPO_ROOT/kmtricks_amd/csrc
for T in 1024; do
  make clean >/dev/null; make -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DKMX_PHASE_PROF -DKMX_ROWS_TPB=$T" >/dev/null 2>&1
  cd ../..; echo "=== TPB=$T"; true
  KMX_MERGE_KERNEL=pivot python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -10 | cut -c1-230; cd kmtricks_amd/csrc
done
