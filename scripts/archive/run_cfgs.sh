# A/B of compile-time geometries: one bench line per -D set given as arguments
cd $GRAFT_REPO_ROOT/kmtricks_amd/csrc
for F in "$@"; do
  make clean >/dev/null; make -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $F" >/dev/null 2>&1
  cd ../..; echo "=== $F"
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(round(r['kernel_ms'],3), 'ms kernel', round(d['ms_per_step'],3), 'ms/step', 'frac', r['frac'], 'rows', d['config']['rows_out_per_step_per_gpu'])"
  cd kmtricks_amd/csrc
done
