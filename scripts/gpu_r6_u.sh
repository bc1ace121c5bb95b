#!/bin/bash
# round 6, call U: the GPU suite as the driver runs it, three times in a row on one box (anything intermittent?)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r6u; mkdir -p $OUT; rm -f $OUT/*
for rep in 1 2 3; do
  ( timeout 900 python -m pytest tests -m gpu -x -q --timeout 600 ) > $OUT/pytest_$rep.log 2>&1; echo "suite run $rep: rc=$? $(tail -1 $OUT/pytest_$rep.log)"
done
python - <<'PY'
import __graft_entry__ as g
g.smoke()
PY
