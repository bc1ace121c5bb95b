"""CPU test of the gapped stage's DP kernel: the piece evaluator of k_ydrop2 (cactus_amd/csrc/mb_ydrop2.h, the source the GPU runs) compiled
for the host against the stand-in HIP header of tests/emu -- one pthread per lane of the wave; DPP moves and scans, readlane, readfirstlane
and ballot exchanged through the wave's slots -- and compared with a plain restatement of SURVEY A.10 ONE_SIDED (one cell at a time in
row-major order): best cell, cells and rows counted, and the alignment read back from the kernel's 4-bit trace codes through its row records;
forward and backward sides that run into the contig ends, Cactus's y-drops (3000: rows inside the first 256 columns; 9400: rows that need
the second group), an N, soft-masked bases; and every side once more cut in two pieces, the second continuing from the first one's exit
snapshot (the relay / continuation format of the gapped stage); the same sides through the four-wave kernel body with the previous row
in an LDS ring (mb_ydrop_lds.h: where rows go that outgrow the one-wave kernel), also in its walls variant (miblast_params.walls: stretches of
the side's own path moved a few columns aside as earlier alignments whose cells are dead); and the relay hand-over check k_verify (mb_verify.h) on the states the
evaluator writes: entry and exit state after the same row are equal, a state moved by one constant and by whole columns / rows is accepted
under the job's offsets, a changed live C or reachable D is rejected, a D that can never matter again may differ; and the traceback kernels (mb_trace.h: a walker per piece, join walks from predicted entries --
every other prediction made wrong on purpose in a second pass --, the stitch) over those chains of pieces give the rule's alignment op for
op.  The emulation is test
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


def test_emulated_relay_hand_over_that_is_accepted_is_exact():
    """The claim the gapped stage's speed rests on: a relay -- a fresh DP started COLD at a cell of the alignment's own
    path, as the stage starts one at a downstream anchor -- whose state after a row equals the upstream piece's state after the same row up
    to one constant (k_verify) evolves like it from there on.  Emulated evaluator and emulated check: whenever the check accepts, the relay
    continued from its snapshot ends at the side's best cell with the side's score (plus the constant) and its traceback is the side's own
    down to the hand-over row; a hand-over may also be rejected (too short a warm-up for a wide window) -- then nothing is claimed -- but
    not all of them."""
    subprocess.run(["make", "-C", EMU_DIR, "emu_ydrop"], check=True, capture_output=True)
    p = subprocess.run([os.path.join(EMU_DIR, "emu_ydrop"), "5", "3", "relay"], capture_output=True, timeout=1500)
    out = p.stdout.decode()
    assert p.returncode == 0, out + p.stderr.decode()
    assert "MISMATCH" not in out and "accepted  ok" in out, out


def test_emulated_hand_over_inside_the_launch_leaves_the_results_alone():
    """Round 5: a piece that reaches its stop row checks its hand-over itself and, rejected, goes on in the same wave
    -- to the relay's next entry snapshot, then to the relay after -- instead of leaving that to a continuation piece in a launch of its
    own.  Emulated launch of a head aimed at relay A aimed at relay B (the relays' blocks first, so that their snapshots are there): as the
    pieces decide for themselves, with the first check of every piece rejected on purpose, and with the first three -- whatever they do,
    following the chain as the host does (k_verify on the hand-over each piece says it ended at) gives the side's best cell, score and
    cell count, and the VerifyJob a piece rewrites is the hand-over it ended at."""
    subprocess.run(["make", "-C", EMU_DIR, "emu_ydrop"], check=True, capture_output=True)
    p = subprocess.run([os.path.join(EMU_DIR, "emu_ydrop"), "9", "1", "inline"], capture_output=True, timeout=1500)
    out = p.stdout.decode()
    assert p.returncode == 0, out + p.stderr.decode()
    assert "MISMATCH" not in out and out.count("concluded  ok") >= 2 and "pass 2" in out, out
