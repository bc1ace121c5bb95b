#!/bin/bash
# More seeds through the host emulation of the kernels (tests/emu) than the CPU suite takes: DP evaluator / LDS body / walls / hand-over check /
# traceback (emu_ydrop), accepted relays (emu_ydrop ... relay), the ungapped kernels in every mode, both seed stages, the grouping of a strand's hits by diagonal in LDS.  No GPU involved; about
# an hour on 8 cores.   usage: bash scripts/emu_fuzz.sh [out_dir]   -> <out_dir>/summary.log (rc per run; 0 = every case identical)
OUT=${1:-/tmp/emufuzz}; mkdir -p "$OUT"; : > "$OUT/summary.log"
cd "$(dirname "$0")/.." || exit 1
make -C tests/emu emu_ydrop emu_ungapped emu_seed_dense emu_seed_batch > /dev/null || exit 1
for s in 21 22 23 24 25; do timeout 3000 ./tests/emu/emu_ydrop $s 6 > "$OUT/ydrop_$s.log" 2>&1; echo "ydrop $s rc=$?" >> "$OUT/summary.log"; done
for s in 31 32 33 34; do timeout 3000 ./tests/emu/emu_ydrop $s 8 relay > "$OUT/relay_$s.log" 2>&1; echo "relay $s rc=$?" >> "$OUT/summary.log"; done
for m in "" ux lane h16; do for s in 41 42 43; do timeout 3000 ./tests/emu/emu_ungapped $s 8 $m > "$OUT/ung_${m}_$s.log" 2>&1; echo "ungapped $m $s rc=$?" >> "$OUT/summary.log"; done; done
for s in 51 52 53 54 55 56; do timeout 3000 ./tests/emu/emu_seed_dense $s 12 > "$OUT/dense_$s.log" 2>&1; echo "dense $s rc=$?" >> "$OUT/summary.log"; done
for s in 71 72 73 74 75 76; do timeout 3000 ./tests/emu/emu_seed_dense $s 28 bin > "$OUT/bin_$s.log" 2>&1; echo "bins $s rc=$?" >> "$OUT/summary.log"; done      # (mb_seed_bin.h: plan, both scatters, both sorters)
for s in 61 62 63; do timeout 3000 ./tests/emu/emu_seed_batch $s 6 > "$OUT/batch_$s.log" 2>&1; echo "batch $s rc=$?" >> "$OUT/summary.log"; done
grep -c "rc=0" "$OUT/summary.log"; grep -v "rc=0" "$OUT/summary.log"
