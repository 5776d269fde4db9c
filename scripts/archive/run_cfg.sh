set -e
cd $GRAFT_REPO_ROOT/kmtricks_amd/csrc
for cfg in "512 2048" "512 4096" "1024 4096"; do
  set -- $cfg
  make clean >/dev/null; make -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DKMX_ROWS_TPB=$1 -DKMX_ROWS_CAP=$2" >/dev/null 2>&1
  cd ../..; echo "=== TPB=$1 CAP=$2"; python -m pytest tests/test_merge_gpu.py -x -q -m gpu 2>&1 | tail -1
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(round(d['value']/1e9,1), 'Gk/s', round(r['kernel_ms'],3), 'ms', round(r['achieved']), 'GB/s')"
  python bench.py --samples 100 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('N=100:', round(d['value']/1e9,1), 'Gk/s', round(r['kernel_ms'],3), 'ms', round(r['achieved']), 'GB/s')"
  cd kmtricks_amd/csrc
done
