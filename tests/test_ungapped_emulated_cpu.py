"""Ungapped extension kernels without a GPU: the product's own kernel sources (cactus_amd/csrc/mb_runs.h -- run heads, anchors of the
HSPs --, mb_ungapped_grp.h -- eight lanes per diagonal run -- and mb_ungapped_ux.h -- the level-synchronous pipeline for dense hit
sets) built for the host against the stand-in
HIP header of tests/emu (one pthread per work-item; DPP, ballot and readlane exchanged through per-wave barriers) and compared with
a sequential restatement of the rule of oracle/lastz_oracle.c:227-262, :508-521 (suppression per diagonal, x-drop both ways,
counters, extent[], HSP records, run-head lists, anchor offsets) on random sequence sets with planted homology, separators, N bases, busy diagonals, extents left
by an earlier q batch and a tiny entry list.  The GPU parity tests proper are tests/test_parity_gpu.py (whole pipeline against the
C oracle, every kernel choice forced in turn).  The emulation is test infrastructure: libmiblast.so has no CPU path."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU = os.path.join(EMU_DIR, "emu_ungapped")


@pytest.fixture(scope="module", autouse=True)
def built():
    subprocess.run(["make", "-C", EMU_DIR, "emu_ungapped"], check=True, capture_output=True)


@pytest.mark.parametrize("mode,seed,cases", [("grp", 5, 2), ("ux", 7, 3), ("ux", 8, 2)])
def test_emulated_ungapped_kernels_match_the_sequential_rule(mode, seed, cases):
    p = subprocess.run([EMU, str(seed), str(cases)] + (["ux"] if mode == "ux" else []), capture_output=True, timeout=900)
    out = p.stdout.decode()
    assert p.returncode == 0, out + p.stderr.decode()
    assert out.count(" ok\n") == cases and "MISMATCH" not in out, out
