#!/bin/bash
# session 65: same-box A/B of the vector body with the gathers of all IPT vectors batched and unpredicated (this tree) against HEAD (tmp_prev/)
# branch-free (this tree): tools/stream_shape_ab.py twice each, interleaved
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s65; mkdir -p $O
for r in 1 2; do
  (cd tmp_prev && timeout -k 10 400 python tools/stream_shape_ab.py > ../$O/prev_$r.txt 2>&1); echo "prev $r exit $?"
  timeout -k 10 400 python tools/stream_shape_ab.py > $O/new_$r.txt 2>&1; echo "new $r exit $?"
done
python - <<'PY'
import re
def parse(p):
    out={}; name=None
    for l in open(p):
        if l[0] not in " \t" and ":" in l and "rows" in l: name=l.split(":")[0]; k=0
        elif l.startswith("   policy") and name:
            pol=l.split()[1]; us=float(l.split(")")[1].split()[0]); out[(name,k,pol)]=us; k+=1
    return out
P=[parse(f"gpurun_out/s65/prev_{r}.txt") for r in (1,2)]; N=[parse(f"gpurun_out/s65/new_{r}.txt") for r in (1,2)]
for key in P[0]:
    if key[1] in (0,) or key[2] in ("2",):
        p=[d.get(key) for d in P]; n=[d.get(key) for d in N]
        print(f"{key[0]:36s} #{key[1]} policy {key[2]}: prev {p}  new {n}")
PY
