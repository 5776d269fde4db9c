#!/bin/bash
# round 6: `kmx pipeline` on 1000 x 5 Mbp (RAM file system), variants of flags / environment over the same inputs; VARIANTS / ENVS as scripts/bench_pipeline.py takes them
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6pipe${TAG:-}
mkdir -p $O
python $GRAFT_REPO_ROOT/scripts/bench_pipeline.py --samples ${SAMPLES:-1000} --genome ${GENOME:-5e6} --partitions 256 --tmp /dev/shm --extra "--hard-min 2 --recurrence-min 2 --static-repart" \
  --variants "${VARIANTS:-}" --env "${ENVS:-}" ${MORE:-} > $O/lines.jsonl 2> $O/err.log
python - <<PY
import json
for l in open("$O/lines.jsonl"):
    if not l.startswith("{"): continue
    d = json.loads(l)
    print({k: d.get(k) for k in ("flags", "env", "count_wall_s", "merge_wall_s", "total_s", "wall_s", "count_s", "read_s", "gpu_workers")})
PY
