"""CPU test of the gapped stage's DP kernel: the piece evaluator of k_ydrop2 (cactus_amd/csrc/mb_ydrop2.h, the source the GPU runs) compiled
for the host against the stand-in HIP header of tests/emu -- one pthread per lane of the wave; DPP moves and scans, readlane, readfirstlane
and ballot exchanged through the wave's slots -- and compared with a plain restatement of SURVEY A.10 ONE_SIDED (one cell at a time in
row-major order): best cell, cells and rows counted, and the alignment read back from the kernel's 4-bit trace codes through its row records;
forward and backward sides that run into the contig ends, Cactus's y-drops (3000: rows inside the first 256 columns; 9400: rows that need
the second group), an N, soft-masked bases; and every side once more cut in two pieces, the second continuing from the first one's exit
snapshot (the relay / continuation format of the gapped stage); and the relay hand-over check k_verify (mb_verify.h) on the states the
evaluator writes: entry and exit state after the same row are equal, a state moved by one constant and by whole columns / rows is accepted
under the job's offsets, a changed live C or reachable D is rejected, a D that can never matter again may differ.  The emulation is test
infrastructure: libmiblast.so has no CPU path."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")


def test_emulated_ydrop2_piece_evaluator_matches_the_rule_cell_by_cell():
    subprocess.run(["make", "-C", EMU_DIR, "emu_ydrop"], check=True, capture_output=True)
    p = subprocess.run([os.path.join(EMU_DIR, "emu_ydrop"), "3", "2"], capture_output=True, timeout=1500)
    out = p.stdout.decode()
    assert p.returncode == 0, out + p.stderr.decode()
    assert out.count(" ok\n") == 2 and "MISMATCH" not in out, out
