#!/bin/bash
# N lastz front-end processes sharing one GPU (the reference's job model: one process per chunk pair)
R=$(cd $(dirname $0)/.. && pwd); cd /tmp; mkdir -p mp && cd mp
python - <<PY
import sys; sys.path.insert(0, "$R")
from cactus_amd import gen
for k in range(8):
    t, q = gen.make_pair(1000000, 42 + k)
    gen.write_fasta("T%d.fa" % k, [("id=simT%d|chr1" % k, t)]); gen.write_fasta("Q%d.fa" % k, [("id=simQ%d|chr1" % k, q)])
PY
ARGS="--format=paf:wfmash --step=1 --ambiguous=iupac,100,100 --ydrop=4000 --hspthresh=2200 --gappedthresh=2400 --queryhspbest=100000"
$R/bin/lastz T0.fa Q0.fa $ARGS > /dev/null   # warm file cache / driver
t0=$(date +%s.%N); for k in 0 1 2 3 4 5 6 7; do $R/bin/lastz "T$k.fa[multiple][nameparse=darkspace]" "Q$k.fa[nameparse=darkspace]" $ARGS > seq$k.paf; done; t1=$(date +%s.%N)
for k in 0 1 2 3 4 5 6 7; do $R/bin/lastz "T$k.fa[multiple][nameparse=darkspace]" "Q$k.fa[nameparse=darkspace]" $ARGS > par$k.paf & done; wait; t2=$(date +%s.%N)
python -c "print(\"8 pairs sequential: %.2f s   8 processes concurrently: %.2f s\" % ($t1 - $t0, $t2 - $t1))"
for k in 0 1 2 3 4 5 6 7; do cmp seq$k.paf par$k.paf || echo DIFF $k; done; wc -l seq0.paf
