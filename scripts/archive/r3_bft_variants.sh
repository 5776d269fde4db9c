# k_merge_bft build variants on one box: merge_bft.o rebuilt with the given macros, libkmx relinked, bench.py --workload bft
cd $GRAFT_REPO_ROOT/kmtricks_amd/csrc
for V in "" "-DKMX_BT_G=8" "-DKMX_BT_G=32" "-DKMX_BT_UNR=6" "-DKMX_BT_UNR=10" "-DKMX_BT_RB=4" "-DKMX_BT_RB=1" "-DKMX_BT_IMG_KB=32" "-DKMX_BT_G=8 -DKMX_BT_UNR=12"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $V -c merge_bft.hip -o merge_bft.o 2>/dev/null || { echo "=== [$V] does not build"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libkmx.so *.o
  echo "=== [$V]"
  (cd ../..; python bench.py --workload bft --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['roofline']['frac'], d['ms_per_step'])")
done
