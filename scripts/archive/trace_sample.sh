# per-stage wall times of the split + count of a few samples (KMX_TRACE lines of libkmx)
T=$(mktemp -d); python - "$T" <<'PY'
import sys, os, numpy as np
T = sys.argv[1]
rng = np.random.default_rng(1); G = 1000000
ref = rng.integers(0, 4, G, dtype=np.uint8); L = 150
with open(f"{T}/in.fof", "w") as fof:
    for s in range(6):
        st = rng.integers(0, G - L, G * 6 // L); reads = np.frombuffer(b"ACTG", np.uint8)[ref[st[:, None] + np.arange(L)[None, :]]]
        lines = np.empty((len(st), L + 4), np.uint8); lines[:, 0] = ord(">"); lines[:, 1] = ord("r"); lines[:, 2] = 10; lines[:, 3:3 + L] = reads; lines[:, 3 + L] = 10
        lines.tofile(f"{T}/S{s}.fa"); fof.write(f"S{s}: {T}/S{s}.fa\n")
PY
KMX_TRACE=1 $GRAFT_REPO_ROOT/kmtricks_amd/kmx pipeline --file $T/in.fof --run-dir $T/run --nb-partitions 256 --static-repart --until count $* 2>&1 | grep "superk_partition\|count_reads\|kmx pipeline" | tail -5
rm -rf $T
