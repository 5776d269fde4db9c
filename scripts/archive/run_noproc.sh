set -e
cd $GRAFT_REPO_ROOT/kmtricks_amd/csrc
make clean >/dev/null; make -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DKMX_PV_NOPROC" >/dev/null 2>&1
cd ../..
KMX_MERGE_KERNEL=pivot python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('NOPROC', round(r['kernel_ms'],3), 'ms', d['config']['rows_out_per_step_per_gpu'])"
