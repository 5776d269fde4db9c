# phase profile of the merge kernel (clock64 per phase, summed over workgroups); extra -D flags in $1,
# KMX_MERGE_KERNEL from the environment
set -e
cd $GRAFT_REPO_ROOT/kmtricks_amd/csrc
make clean >/dev/null; make -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DKMX_PHASE_PROF $1" >/dev/null 2>&1
cd ../..
python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -24 | cut -c1-260
