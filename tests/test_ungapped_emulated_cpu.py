"""Ungapped extension kernels without a GPU: the product's own kernel sources (cactus_amd/csrc/mb_runs.h -- run heads, anchors of the
HSPs --, mb_ungapped_lane.h -- a diagonal run per lane, and the wave-per-run kernel every choice hands the busy diagonals to --,
mb_ungapped_grp.h -- eight lanes per diagonal run -- and mb_ungapped_ux.h -- the level-synchronous pipeline for dense hit
sets --, mb_hash16.h -- lastz's 16-bit diagonal hash as a mode: every hit extended, the rule per hash class in generation order, on targets long
enough for diagonals 65536 apart to share a class) built for the host against the stand-in
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


@pytest.mark.parametrize("mode,seed,cases", [("grp", 5, 2), ("ux", 7, 3), ("ux", 8, 2), ("lane", 9, 3), ("h16", 4, 4)])
def test_emulated_ungapped_kernels_match_the_sequential_rule(mode, seed, cases):
    p = subprocess.run([EMU, str(seed), str(cases)] + ([mode] if mode in ("ux", "lane", "h16") else []), capture_output=True, timeout=900)
    out = p.stdout.decode()
    assert p.returncode == 0, out + p.stderr.decode()
    assert out.count(" ok\n") == cases and "MISMATCH" not in out, out


def test_emulated_packed_windows_and_scrambled_extent_slots_match_the_sequential_rule():
    """Round 6: level 1 and 2 of the level-synchronous pipeline from the PACKED strands (k_ux_extend_pk: 12-byte records of 32 bases, code bytes
    through a table in LDS; windows with an N, a separator or an end of the set take the byte loads) and extent[] kept in the order of the
    scrambled keys (UnitTab::ext_mul: a strand in several q batches) -- same HSPs, counters and extents as the sequential rule."""
    p = subprocess.run([EMU, "3", "16", "ux"], capture_output=True, timeout=1800)
    out = p.stdout.decode()
    assert p.returncode == 0, out + p.stderr.decode()
    assert out.count(" ok\n") == 16 and "MISMATCH" not in out, out
    assert out.count("level 1 from the packed strands") >= 3 and out.count("extent slots in the order of the scrambled keys") >= 1, out


def test_emulated_batched_seed_stage_matches_the_rule():
    """cactus_amd/csrc/mb_seed_batch.h on the host: the sparse seed tables of several targets (bitmap + rank directory + CSR) hold
    exactly the positions SURVEY A.3 indexes, and the seed search over all (pair, strand) units of a call writes, unit by unit in
    q order, exactly the hits of SURVEY A.4 (every word variant, every position of its bucket)."""
    subprocess.run(["make", "-C", EMU_DIR, "emu_seed_batch"], check=True, capture_output=True)
    p = subprocess.run([os.path.join(EMU_DIR, "emu_seed_batch"), "11", "3"], capture_output=True, timeout=900)
    out = p.stdout.decode()
    assert p.returncode == 0, out + p.stderr.decode()
    assert out.count(" ok\n") == 3 and "MISMATCH" not in out, out


def test_emulated_dense_seed_stage_matches_the_rule():
    """cactus_amd/csrc/mb_seed_dense.h on the host (tests/emu/emu_seed_dense.cpp): a strand packed to 2 + 1 bits per base holds the
    bases and the mask a plain loop computes; the target's seed words taken from the packed form are window_word's; the q-ordered
    one-pass search (k_seed_hits: a tile lists its hits into scratch reserved with one atomic add; k_seed_keys: a block per tile writes
    the keys at the tile's place in q order) gives, query position by query position, exactly the hits of SURVEY A.4 -- thirteen word
    variants or one, packed or byte-code query, scrambled diagonals (undone by k_keys_unhash) or plain ones --, and a key buffer that
    is too small gets the total and no write past its end; and of mb_seed_index.h the strand in q batches (k_seed_count, scan, k_seed_fill) and
    the one-pass search without q order (k_seed_search)."""
    subprocess.run(["make", "-C", EMU_DIR, "emu_seed_dense"], check=True, capture_output=True)
    p = subprocess.run([os.path.join(EMU_DIR, "emu_seed_dense"), "7", "10"], capture_output=True, timeout=900)
    out = p.stdout.decode()
    assert p.returncode == 0, out + p.stderr.decode()
    assert out.count(" ok\n") == 10 and "MISMATCH" not in out, out


def test_emulated_grouping_by_diagonal_in_lds_equals_sort_and_unscramble():
    """cactus_amd/csrc/mb_seed_bin.h on the host: the plan (bins from the key count the device reads, their places, the largest bin, the bins
    beyond the small sorter; nothing planned for keys that did not fit their buffer), the scatter (every bin filled exactly) and the two
    LDS sorters -- buckets, ranks, unscrambling, guard words around the output -- give the array std::sort by (scrambled diagonal, q) and the
    unscrambling give: one bin and hundreds, a dozen keys and tens of thousands, diagonals with thousands of hits (the large sorter), plain
    and scrambled diagonals."""
    subprocess.run(["make", "-C", EMU_DIR, "emu_seed_dense"], check=True, capture_output=True)
    p = subprocess.run([os.path.join(EMU_DIR, "emu_seed_dense"), "2", "11", "bin"], capture_output=True, timeout=900)
    out = p.stdout.decode()
    assert p.returncode == 0, out + p.stderr.decode()
    assert out.count(" ok\n") == 11 and "MISMATCH" not in out, out
    assert "1 beyond the small sorter" in out, "no case reached the large sorter"


@pytest.mark.skipif(not os.environ.get("MIBLAST_SLOW_TESTS"), reason="three minutes of emulated scan over the 2^24 + 1 bucket counts: MIBLAST_SLOW_TESTS=1")
def test_emulated_dense_seed_table_through_its_kernels():
    """mb_seed_index.h on the host: index words from the byte codes (= the packed ones), the three-launch exclusive scan of the bucket
    counts with the occupancy bitmap and the counts left zeroed, the scatter, the cleared cursors -- against the plain table."""
    subprocess.run(["make", "-C", EMU_DIR, "emu_seed_dense"], check=True, capture_output=True)
    p = subprocess.run([os.path.join(EMU_DIR, "emu_seed_dense"), "7", "1", "table"], capture_output=True, timeout=1800)
    out = p.stdout.decode()
    assert p.returncode == 0 and out.count(" ok\n") == 1 and "MISMATCH" not in out, out + p.stderr.decode()


def test_emulated_set_kernels_match_plain_loops():
    """cactus_amd/csrc/mb_sets.h on the host (tests/emu/emu_sets.cpp): the '-' strands of a call's query sets, and outgroup trimming
    between two calls -- interval marks, edges of the uncovered stretches, the gathered device image of the new set with its contig
    tables -- all sets of a call in one launch, against plain loops over every set."""
    subprocess.run(["make", "-C", EMU_DIR, "emu_sets"], check=True, capture_output=True)
    p = subprocess.run([os.path.join(EMU_DIR, "emu_sets"), "5", "60"], capture_output=True, timeout=900)
    assert p.returncode == 0, p.stdout.decode()[-2000:]
    assert b"60 cases, 0 differences" in p.stdout
