"""Chaining stage without a GPU: the product's own sources (cactus_amd/csrc/mp_chain.cpp, mp_kernels.hip, mp_paffy_main.cpp) built
for the host against the stand-in HIP headers of tests/emu (one pthread per work-item, barriers for __syncthreads and the wave
shuffles) and compared with the oracle byte for byte.  This covers the host orchestration (sort passes, group bounds, chain
peeling, op splicing) and the kernels' logic in the CPU suite; the GPU parity tests proper are tests/test_zz_chain_gpu.py.
The emulation is test infrastructure: it is not part of libmiblast.so, which has no CPU path."""
import os
import subprocess

import pytest

from tests import pyref_paffy as ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle", "oracle_paffy")
EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU = os.path.join(EMU_DIR, "emu_paffy")
CHAIN_ARGS = ["--maxGapLength", "1000000", "--chainGapOpen", "5000", "--chainGapExtend", "1", "--trimFraction", "1.0"]
TIGHT_ARGS = ["--maxGapLength", "3000", "--chainGapOpen", "100", "--chainGapExtend", "3", "--trimFraction", "0.25"]


@pytest.fixture(scope="module", autouse=True)
def built():
    subprocess.run(["make", "-C", EMU_DIR], check=True, capture_output=True)


def run(exe, cmd, text, *args):
    # 4 waves per workgroup of the chain DP instead of 16: fewer pthreads per emulated group, same code path
    p = subprocess.run([exe, cmd, *args], input=text.encode(), capture_output=True, env=dict(os.environ, MIPAF_CHAIN_THREADS="256"))
    assert p.returncode == 0, p.stderr.decode()
    return p.stdout.decode()


def both_ways(seed, **kw):
    text = ref.random_paf(seed, **kw)
    return text + ref.dump(ref.invert(ref.parse(text)))


@pytest.mark.parametrize("seed", range(2))
def test_emulated_kernels_match_the_oracle_step_by_step(seed):
    text = both_ways(seed, n_series=4 + seed, noise=8, contig_len=60_000 if seed % 2 else 200_000)
    chained = run(ORACLE, "chain", text, *CHAIN_ARGS)
    assert run(EMU, "chain", text, *CHAIN_ARGS) == chained
    assert run(EMU, "chain", text, *TIGHT_ARGS) == run(ORACLE, "chain", text, *TIGHT_ARGS)
    if seed == 0:                                            # one wave / sixteen waves per workgroup
        for threads in ("64", "1024"):
            p = subprocess.run([EMU, "chain", *CHAIN_ARGS], input=text.encode(), capture_output=True, env=dict(os.environ, MIPAF_CHAIN_THREADS=threads))
            assert p.returncode == 0 and p.stdout.decode() == chained, threads
    tiled = run(ORACLE, "tile", chained)
    assert run(EMU, "tile", chained) == tiled                                     # sort-based levelling
    assert run(EMU, "tile", chained, "--mipaf-hist-bins", "2") == tiled          # counter walk, every level above 1 through the bisection
    if seed == 1:                                            # pile-up guard: falls back to the walk
        p = subprocess.run([EMU, "tile"], input=chained.encode(), capture_output=True, env=dict(os.environ, MIPAF_TILE_MAX_PIECES="20"))
        assert p.returncode == 0 and p.stdout.decode() == tiled
    for x in ("0.2", "0.97", "1"):
        assert run(EMU, "trim", tiled, "--trimIdentity", x) == run(ORACLE, "trim", tiled, "--trimIdentity", x), x


def test_emulated_deep_tiling_and_long_groups():
    # one query/target pair, many overlapping alignments: long groups, equal scores, levels >= 3
    text = both_ways(77, n_series=9, per_series=(15, 30), n_q=1, n_t=1, contig_len=300_000, noise=80)
    assert len(text.splitlines()) > 350
    chained = run(ORACLE, "chain", text, *CHAIN_ARGS)
    assert run(EMU, "chain", text, *CHAIN_ARGS) == chained
    tiled = run(ORACLE, "tile", chained)
    assert max(int(l.split("tl:i:")[1].split("\t")[0]) for l in tiled.splitlines()) >= 3
    assert run(EMU, "tile", chained) == tiled
    assert run(EMU, "tile", chained, "--mipaf-hist-bins", "3") == tiled


def test_emulated_edge_cases():
    assert run(EMU, "chain", "") == "" and run(EMU, "tile", "") == "" and run(EMU, "trim", "", "--trimIdentity", "0.2") == ""
    one = "q\t100\t10\t20\t-\tt\t200\t30\t40\t10\t10\t255\n"
    for cmd, args in (("chain", CHAIN_ARGS), ("tile", []), ("trim", ["--trimIdentity", "0.2"])):
        assert run(EMU, cmd, one, *args) == run(ORACLE, cmd, one, *args)
    # columns 10/11 that disagree with the cigar: R-R3 rewrites them for every kept record, trimmed or not
    odd = "q\t1000\t100\t200\t+\tt\t2000\t300\t400\t50\t70\t255\tAS:i:9000\tcg:Z:100=\n" \
          "q\t1000\t300\t420\t-\tt\t2000\t500\t600\t7\t9\t255\tAS:i:500\tcg:Z:40=20I10X50=\n"
    got = run(EMU, "trim", odd, "--trimIdentity", "0.2")
    assert got == run(ORACLE, "trim", odd, "--trimIdentity", "0.2") and "\t100\t100\t255" in got
    p = subprocess.run([EMU, "tile"], input=b"q\t100\t10\t20\t+\tt\t200\t30\t40\t10\t10\t255\tcg:Z:11=\n", capture_output=True)
    assert p.returncode == 1 and b"do not agree" in p.stderr and p.stdout == b""


def test_emulated_pipeline_reproduces_the_committed_chain_fixtures():
    g = lambda name: open(os.path.join(ROOT, "tests", "golden", f"chain_{name}.paf")).read()      # noqa: E731
    assert run(EMU, "chain", g("input"), *CHAIN_ARGS) == g("chain")
    assert run(EMU, "tile", g("chain")) == g("tile")
    assert run(EMU, "trim", g("tile"), "--trimIdentity", "0.2") == g("trim")
    assert run(EMU, "filter", g("trim"), "--maxTileLevel", "1") == g("primary")
    assert run(EMU, "chain", g("primary"), *CHAIN_ARGS) == g("rechain")
    assert run(EMU, "filter", g("rechain"), "--minChainScore", "10000") == g("output")


@pytest.mark.parametrize("secondary", ["0", "1"])
def test_chain_alignments_job_function_over_the_emulated_front_end(tmp_path, monkeypatch, secondary):
    """chain_alignments / chain_tile_trim_filter_one_contig (the mirrors of local_alignment.py:607-727) as the reference's caller runs
    them, with `paffy` on PATH being the host build of the product's sources: merged input, inverted copies, split by query contig,
    piped per-part jobs (both outputSecondaryAlignments modes), merge.  The GPU suite runs the same test with bin/paffy."""
    import xml.etree.ElementTree as ET
    from cactus_amd.paf import local_alignment as la
    from cactus_amd.shared import common
    from localjob import LocalJob
    (tmp_path / "bin").mkdir()
    os.symlink(EMU, tmp_path / "bin" / "paffy")
    monkeypatch.setattr(common, "BIN_DIR", str(tmp_path / "bin"))
    monkeypatch.setenv("MIPAF_CHAIN_THREADS", "256")
    monkeypatch.delenv("MIBLAST_INPROCESS", raising=False)
    params = ET.parse(os.path.join(ROOT, "cactus_amd", "blast_config.xml")).getroot()
    blast = params.find("blast")
    blast.attrib["outputSecondaryAlignments"] = secondary
    blast.attrib["chainSplitMinSize"] = "1000"
    blast.attrib["chainContigGroupSize"] = "150000"
    job = LocalJob()
    parts = [ref.random_paf(51, n_series=4, noise=6), ref.random_paf(52, n_series=4, noise=6)]
    ids = []
    for k, text in enumerate(parts):
        path = os.path.join(job.fileStore.getLocalTempDir(), f"{k}.paf")
        open(path, "w").write(text)
        ids.append(job.fileStore.writeGlobalFile(path))
    out = open(str(la.chain_alignments(job, ids, ["a", "b"], "Anc0", params))).read()

    def oracle_job(text):
        filt = run(ORACLE, "filter", run(ORACLE, "trim", run(ORACLE, "tile", run(ORACLE, "chain", text, *CHAIN_ARGS)), "--trimIdentity", "0.2"), "--maxTileLevel", "1")
        rechained = run(ORACLE, "chain", filt, *CHAIN_ARGS)
        if secondary == "0":
            return run(ORACLE, "filter", rechained, "--minChainScore", "10000")
        demoted = run(ORACLE, "filter", rechained, "--invert", "--minChainScore", "10000").replace("tp:A:P", "tp:A:S").replace("tl:i:1", "tl:i:2")
        return run(ORACLE, "filter", filt, "--maxTileLevel", "1", "--invert") + run(ORACLE, "filter", rechained, "--minChainScore", "10000") + demoted

    merged = "".join(parts)
    merged += run(ORACLE, "invert", merged)
    groups, seen, acc = [[]], {}, 0                          # R-S1: parts close at >= 150 kb of query sequence, first appearance order
    for line in merged.splitlines(keepends=True):
        c = line.split("\t")
        if c[0] not in seen:
            seen[c[0]] = len(groups) - 1
            acc += int(c[1])
            if acc >= 150000:
                groups.append([]); acc = 0
        groups[seen[c[0]]].append(line)
    assert out == "".join(oracle_job("".join(g)) for g in groups if g) and len([g for g in groups if g]) >= 2
