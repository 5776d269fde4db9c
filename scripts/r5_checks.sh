#!/bin/bash
# round 5: random sweeps of split + count at k = 64 ... 127 against the oracle; the byte-wide side store below 512 lists
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5c; mkdir -p $O
timeout 900 python scripts/fuzz_count.py --kmers wide --cases 30 --seed 5 --scale 2 2>&1 | tail -2 > $O/fuzz_wide.txt
timeout 600 python scripts/fuzz_count.py --kmers all --cases 30 --seed 6 2>&1 | tail -2 >> $O/fuzz_wide.txt
for N in 256 320 400; do
  KMX_MERGE_KERNEL=cols timeout 600 python scripts/r5_grid.py --samples $N --steps 10 --settings "x=1;KMX_DENSE_NARROW=1" 2>/dev/null | sed "s/^/N=$N /" >> $O/narrow_mid.txt
done
cat $O/fuzz_wide.txt $O/narrow_mid.txt
