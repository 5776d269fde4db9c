# phase profile (clock64 per phase, -DKMX_PHASE_PROF build made by scripts/dev/build_variant.sh prof) of k_merge_cols / k_cols_sparse:
# bench.py runs the steps with the rows where the kernels leave them first, then in file order -- two dumps per workload
for wl in ${WLS:-count pa63}; do
  echo "== $wl $EXTRA"
  KMX_LIB=$GRAFT_REPO_ROOT/kmtricks_amd/libkmx_prof.so python bench.py --workload $wl --steps 1 --warmup 1 --no-cpu-baseline --no-whole-job $EXTRA 2>&1 | grep -E "^\[cols\]|^\[sparse\]" > /tmp/ph.txt
  python - <<'PY'
import re, collections
# every result free dumps both tables: sum the dumps of the arena steps (first half) and of the file-order steps (second half)
L = [l for l in open("/tmp/ph.txt") if l.startswith("[sparse]")]
n = 8; dumps = [L[i:i + n] for i in range(0, len(L), n)]
half = len(dumps) // 2
for name, ds in (("rows where the kernels leave them", dumps[:half]), ("rows in file order", dumps[half:])):
    acc = collections.OrderedDict()
    for d in ds:
        for l in d:
            m = re.match(r"\[sparse\] (.+?)\s+[\d.]+%\s+(\d+)", l)
            acc[m.group(1)] = acc.get(m.group(1), 0) + int(m.group(2))
    tot = sum(acc.values()) or 1
    print("--", name, "(%d dumps)" % len(ds))
    for k, v in acc.items(): print("   %-36s %6.2f%% %12d" % (k, 100.0 * v / tot, v // max(1, len(ds))))
PY
done
