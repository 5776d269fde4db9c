# round 2: knobs of the default bench (lists from the count stage) -- KMX_ITEMS_PER_SLOT = work items per resident workgroup slot
cd $GRAFT_REPO_ROOT
for L in counted random; do for ips in 6 8 12 16 24; do echo -n "$L KMX_ITEMS_PER_SLOT=$ips: "; KMX_ITEMS_PER_SLOT=$ips python bench.py --no-cpu-baseline --lists $L 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))"; done; done
