#!/bin/bash
# tests/golden/fuzz_chunk_r06.json: the ORACLE's answers (md5 of the PAF, twelve counters, record counts) of chunk-scale cases of the randomised
# differential run -- scripts/gpu_fuzz.py makes case k from seed k alone (structure, sizes 1.7 - 2.6 Mb, option set), here without a GPU
# (FUZZ_WRITE_DIGESTS), several processes side by side; ~ 40 CPU-minutes on 8 cores.  The GPU side: scripts/gpu_r6_fuzz3.sh (all cases) and
# tests/test_parity_gpu.py::test_chunk_scale_random_cases_against_committed_oracle_digests (four of them).
#   bash tests/golden/make_fuzz_chunk_digests.sh   -> tests/golden/fuzz_chunk_r06.json
cd "$(dirname "$0")/../.." || exit 1
TMP=$(mktemp -d)
export FUZZ_NMIN=1700000 FUZZ_NMAX=2600000
for i in 0 1 2 3 4 5; do FUZZ_WRITE_DIGESTS=$TMP/d_$i.json python scripts/gpu_fuzz.py 2 $((22000 + 2 * i)) > $TMP/d_$i.log 2>&1 & done
wait
for i in 0 1 2 3 4 5 6; do FUZZ_WRITE_DIGESTS=$TMP/e_$i.json python scripts/gpu_fuzz.py 6 $((23000 + 6 * i)) > $TMP/e_$i.log 2>&1 & done
wait
export FUZZ_NMIN=4000000 FUZZ_NMAX=6000000          # -> fuzz_chunk_r06_large.json (20 minutes more)
for i in 0 1 2 3 4 5 6; do FUZZ_WRITE_DIGESTS=$TMP/L_$i.json python scripts/gpu_fuzz.py 1 $((24000 + i)) > $TMP/L_$i.log 2>&1 & done
wait
python - "$TMP" <<'PY'
import glob, json, sys
for pat, out in (("/[de]_*.json", "tests/golden/fuzz_chunk_r06.json"), ("/L_*.json", "tests/golden/fuzz_chunk_r06_large.json")):
    m = {}
    for f in sorted(glob.glob(sys.argv[1] + pat)):
        m.update(json.load(open(f)))
    json.dump(m, open(out, "w"), indent=1, sort_keys=True)
    print(out, len(m), "cases")
PY
rm -rf "$TMP"
