"""GPU parity tests of the chaining stage: libmiblast's chain / tile / trim kernels through the C ABI
(include/mipaf.h) against oracle/oracle_paffy, byte for byte, step by step and as the whole
chain_tile_trim_filter_one_contig job (/root/reference/src/cactus/paf/local_alignment.py:660-727).
PARITY UNPINNED: paffy is an absent submodule of the reference; the rules are DESIGN.md section 11."""
import os
import subprocess

import pytest

from cactus_amd import gen, miblast, mipaf
from cactus_amd.shared.common import BIN_DIR
from tests import pyref_paffy as ref

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle", "oracle_paffy")
PAFFY = os.path.join(BIN_DIR, "paffy")
CHAIN_ARGS = ["--maxGapLength", "1000000", "--chainGapOpen", "5000", "--chainGapExtend", "1", "--trimFraction", "1.0"]      # xml:108-111
TIGHT_ARGS = ["--maxGapLength", "3000", "--chainGapOpen", "100", "--chainGapExtend", "3", "--trimFraction", "0.25"]


def oracle(cmd, text, *args):
    p = subprocess.run([ORACLE, cmd, *args], input=text.encode(), capture_output=True)
    assert p.returncode == 0, p.stderr.decode()
    return p.stdout.decode()


def oracle_job(text, secondary=False):
    """chain_tile_trim_filter_one_contig with the oracle in place of paffy"""
    filt = oracle("filter", oracle("trim", oracle("tile", oracle("chain", text, *CHAIN_ARGS)), "--trimIdentity", "0.2"), "--maxTileLevel", "1")
    rechained = oracle("chain", filt, *CHAIN_ARGS)
    if not secondary:
        return oracle("filter", rechained, "--minChainScore", "10000")
    out = oracle("filter", filt, "--maxTileLevel", "1", "--invert")
    out += oracle("filter", rechained, "--minChainScore", "10000")
    demoted = oracle("filter", rechained, "--invert", "--minChainScore", "10000")
    return out + demoted.replace("tp:A:P", "tp:A:S").replace("tl:i:1", "tl:i:2")


def both_ways(seed, **kw):
    text = ref.random_paf(seed, **kw)
    return text + ref.dump(ref.invert(ref.parse(text)))


@pytest.fixture(scope="module")
def ctx():
    c = miblast.Context(0)
    yield c


TIGHT = mipaf.default_chain_params(max_gap_length=3000, gap_open=100, gap_extend=3, trim_fraction=0.25)


@pytest.mark.parametrize("seed", range(8))
def test_each_sub_command_matches_the_oracle(ctx, seed):
    text = both_ways(seed, n_series=5 + seed, noise=10 + 3 * seed, contig_len=60_000 if seed % 2 else 200_000)
    chained = mipaf.PafSet.from_text(text).chain(ctx).text()
    assert chained == oracle("chain", text, *CHAIN_ARGS)
    assert mipaf.PafSet.from_text(text).chain(ctx, TIGHT).text() == oracle("chain", text, *TIGHT_ARGS)
    tiled = mipaf.PafSet.from_text(chained).tile(ctx).text()                                   # sort-based levelling
    assert tiled == oracle("tile", chained)
    # the counter walk (what a pile-up falls back to); a 2-bin histogram sends every level above 1 through its bisection
    assert mipaf.PafSet.from_text(chained).tile(ctx, hist_bins=4096).text() == tiled
    assert mipaf.PafSet.from_text(chained).tile(ctx, hist_bins=2).text() == tiled
    assert mipaf.PafSet.from_text(text).tile(ctx).text() == oracle("tile", text)                 # no chain scores: AS decides
    for x in ("0.2", "0.5", "0.97", "0", "1"):
        assert mipaf.PafSet.from_text(tiled).trim(ctx, x).text() == oracle("trim", tiled, "--trimIdentity", x), x
    # op ranges that no record owns any more (filtered records, ops cut by an earlier trim) lie between the live ones
    twice = mipaf.PafSet.from_text(tiled).filter(max_tile_level=1).trim(ctx, "0.5").trim(ctx, "0.97").text()
    assert twice == oracle("trim", oracle("trim", oracle("filter", tiled, "--maxTileLevel", "1"), "--trimIdentity", "0.5"), "--trimIdentity", "0.97")


@pytest.mark.parametrize("seed", [21, 22, 23])
@pytest.mark.parametrize("secondary", [False, True])
def test_whole_job_matches_the_oracle_pipeline(ctx, seed, secondary):
    text = both_ways(seed, n_series=12, per_series=(3, 20), noise=40)
    s = mipaf.PafSet.from_text(text).chain_tile_trim_filter(ctx, None, "0.2", 10000, output_secondary=secondary)
    want = oracle_job(text, secondary)
    assert s.text() == want and len(want.splitlines()) > 10
    assert s.stats["records"] == len(want.splitlines()) and s.stats["t_chain_dp_ms"] > 0 and s.stats["t_tile_ms"] > 0


def test_front_end_pipes_like_cactus_call(tmp_path):
    # the composite piped call of local_alignment.py:684-691, with bin/paffy on both sides of every pipe
    text = both_ways(31, n_series=10, noise=30)
    src = tmp_path / "input.paf"
    src.write_text(text)
    chain = f"{PAFFY} chain " + " ".join(CHAIN_ARGS) + " --logLevel INFO"
    cmd = (f"{chain} --inputFile {src} | {PAFFY} tile --logLevel INFO | {PAFFY} trim --trimIdentity 0.2 | {PAFFY} filter --maxTileLevel 1 | "
           f"{chain} | {PAFFY} filter --minChainScore 10000")
    p = subprocess.run(["bash", "-o", "pipefail", "-c", cmd], capture_output=True)
    assert p.returncode == 0, p.stderr.decode()
    assert p.stdout.decode() == oracle_job(text)


def test_chaining_the_blast_output_of_a_synthetic_pair(ctx):
    # blast phase -> chaining stage, both on the GPU; the oracle chains the same PAF
    t, q = gen.make_pair(150_000, 7)
    pm = miblast.params_from_args("--step=1 --ambiguous=iupac,100,100 --ydrop=4000 --hspthresh=2200 --gappedthresh=2400".split())
    T = ctx.seqset_from_fasta_bytes(gen.fasta_bytes([("id=simT|chr1", t)]))
    Q = ctx.seqset_from_fasta_bytes(gen.fasta_bytes([("id=simQ|chr1", q)]))
    paf = ctx.align(T, Q, pm, details=False).paf.decode()
    assert len(paf.splitlines()) >= 3
    text = paf + mipaf.PafSet.from_text(paf).invert().text()                                  # local_alignment.py:620-626
    got = mipaf.PafSet.from_text(text).chain_tile_trim_filter(ctx, None, "0.2", 10000).text()
    assert got == oracle_job(text) and got


def test_large_groups_windows_and_ties(ctx):
    # thousands of alignments in a few (query, target, strand) groups: the predecessor window, equal scores, deep tiling
    text = both_ways(41, n_series=60, per_series=(20, 60), n_q=1, n_t=2, contig_len=3_000_000, noise=600)
    n = len(text.splitlines())
    assert n > 4000
    for params, args in ((None, CHAIN_ARGS), (TIGHT, TIGHT_ARGS)):
        s = mipaf.PafSet.from_text(text).chain(ctx, params)
        assert s.text() == oracle("chain", text, *args)
        assert s.stats["groups"] <= 8 and s.stats["records"] == n
    chained = oracle("chain", text, *CHAIN_ARGS)
    tiled = mipaf.PafSet.from_text(chained).tile(ctx).text()
    assert tiled == oracle("tile", chained)
    assert mipaf.PafSet.from_text(chained).tile(ctx, hist_bins=4096).text() == tiled and mipaf.PafSet.from_text(chained).tile(ctx, hist_bins=3).text() == tiled
    assert max(int(l.split("tl:i:")[1].split("\t")[0]) for l in tiled.splitlines()) >= 3
    assert mipaf.PafSet.from_text(tiled).trim(ctx, "0.2").text() == oracle("trim", tiled, "--trimIdentity", "0.2")
    assert mipaf.PafSet.from_text(text).chain_tile_trim_filter(ctx, None, "0.2", 10000).text() == oracle_job(text)


@pytest.mark.parametrize("threads", ["64", "256", "512"])
def test_chain_dp_workgroup_size_does_not_change_a_byte(ctx, monkeypatch, threads):
    text = both_ways(43, n_series=30, per_series=(10, 40), n_q=1, n_t=1, contig_len=2_000_000, noise=300)
    monkeypatch.setenv("MIPAF_CHAIN_THREADS", threads)
    assert mipaf.PafSet.from_text(text).chain(ctx).text() == oracle("chain", text, *CHAIN_ARGS)
    assert mipaf.PafSet.from_text(text).chain(ctx, TIGHT).text() == oracle("chain", text, *TIGHT_ARGS)


def test_trim_of_a_few_very_long_cigars(ctx):
    # whole-chunk alignments: a handful of records with tens of thousands of ops each (one lane per op, not per record)
    text = both_ways(61, n_series=3, per_series=(1, 2), n_q=1, n_t=1, contig_len=40_000_000, noise=0, ragged=True)
    long_text = "".join(l for l in text.splitlines(keepends=True))
    import re
    def stretch(line):                                      # repeat the cigar 300 times and fix the coordinates
        c = line.rstrip("\n").split("\t")
        cg = [x for x in c if x.startswith("cg:Z:")]
        if not cg:
            return line
        ops = re.findall(r"(\d+)([=XID])", cg[0][5:]) * 300
        qspan = sum(int(n) for n, o in ops if o != "D"); tspan = sum(int(n) for n, o in ops if o != "I")
        c[1] = c[6] = "40000000"
        c[2], c[3] = "1000", str(1000 + qspan)
        c[7], c[8] = "2000", str(2000 + tspan)
        c[9] = str(sum(int(n) for n, o in ops if o == "="))
        c[10] = str(sum(int(n) for n, o in ops))
        c[c.index(cg[0])] = "cg:Z:" + "".join(n + o for n, o in ops)
        return "\t".join(c) + "\n"
    long_text = "".join(stretch(l) for l in long_text.splitlines(keepends=True))
    for x in ("0.2", "0.9", "1"):
        assert mipaf.PafSet.from_text(long_text).trim(ctx, x).text() == oracle("trim", long_text, "--trimIdentity", x), x
    tiled = oracle("tile", long_text)
    assert mipaf.PafSet.from_text(long_text).tile(ctx).text() == tiled


def test_pile_up_falls_back_to_the_counter_walk(ctx, monkeypatch):
    text = both_ways(9, n_series=8, noise=30)
    chained = oracle("chain", text, *CHAIN_ARGS)
    monkeypatch.setenv("MIPAF_TILE_MAX_PIECES", "50")
    s = mipaf.PafSet.from_text(chained).tile(ctx)
    assert s.text() == oracle("tile", chained)


def test_edge_cases(ctx):
    assert mipaf.PafSet.from_text("").chain(ctx).tile(ctx).trim(ctx, "0.2").text() == ""
    one = "q\t100\t10\t20\t-\tt\t200\t30\t40\t10\t10\t255\n"                                    # no AS, no cigar
    assert mipaf.PafSet.from_text(one).chain(ctx).text() == oracle("chain", one, *CHAIN_ARGS)
    assert mipaf.PafSet.from_text(one).tile(ctx).trim(ctx, "0.2").text() == oracle("trim", oracle("tile", one), "--trimIdentity", "0.2")
    # a cigar that does not walk its intervals is refused (the contract caf asserts later)
    with pytest.raises(miblast.MiblastError):
        mipaf.PafSet.from_text("q\t100\t10\t20\t+\tt\t200\t30\t40\t10\t10\t255\tcg:Z:11=\n").tile(ctx)
    with pytest.raises(miblast.MiblastError):
        mipaf.PafSet.from_text(one).trim(ctx, "1.5")
    # same input, same bytes
    text = both_ways(5)
    assert mipaf.PafSet.from_text(text).chain_tile_trim_filter(ctx, None, "0.2", 10000).text() == \
        mipaf.PafSet.from_text(text).chain_tile_trim_filter(ctx, None, "0.2", 10000).text()


@pytest.mark.parametrize("inprocess", ["0", "1"])
@pytest.mark.parametrize("secondary", ["0", "1"])
def test_chain_alignments_job_function(monkeypatch, inprocess, secondary):
    """chain_alignments (local_alignment.py:607-657) as the reference's caller would run it: two PAF files in, inverted copies
    added, split by query contig (threshold lowered so the split path runs), per-contig jobs, merged output."""
    import xml.etree.ElementTree as ET
    from cactus_amd.paf import local_alignment as la
    from localjob import LocalJob
    monkeypatch.setenv("MIBLAST_INPROCESS", inprocess)
    params = ET.parse(os.path.join(ROOT, "cactus_amd", "blast_config.xml")).getroot()
    blast = params.find("blast")
    blast.attrib["outputSecondaryAlignments"] = secondary
    blast.attrib["chainSplitMinSize"] = "1000"
    blast.attrib["chainContigGroupSize"] = "150000"
    job = LocalJob()
    parts = [ref.random_paf(51, n_series=10, noise=20), ref.random_paf(52, n_series=10, noise=20)]
    ids = []
    for k, text in enumerate(parts):
        path = os.path.join(job.fileStore.getLocalTempDir(), f"{k}.paf")
        open(path, "w").write(text)
        ids.append(job.fileStore.writeGlobalFile(path))
    out = open(str(la.chain_alignments(job, ids, ["a", "b"], "Anc0", params))).read()
    merged = "".join(parts)
    merged += oracle("invert", merged)
    # the split: query sequences in order of first appearance, a part closes at >= 150 kb of sequence (R-S1)
    groups, seen, acc = [[]], {}, 0
    for line in merged.splitlines(keepends=True):
        c = line.split("\t")
        if c[0] not in seen:
            seen[c[0]] = len(groups) - 1
            acc += int(c[1])
            if acc >= 150000:
                groups.append([]); acc = 0
        groups[seen[c[0]]].append(line)
    want = "".join(oracle_job("".join(g), secondary == "1") for g in groups if g)
    assert out == want and len(groups) >= 2


def test_committed_chain_fixtures(ctx):
    """tests/golden/chain_*.paf (written by tests/golden/make_chain_golden.py from the oracle): every step and the whole job."""
    g = lambda name: open(os.path.join(ROOT, "tests", "golden", f"chain_{name}.paf")).read()      # noqa: E731
    assert mipaf.PafSet.from_text(g("input")).chain(ctx).text() == g("chain")
    assert mipaf.PafSet.from_text(g("chain")).tile(ctx).text() == g("tile")
    assert mipaf.PafSet.from_text(g("chain")).tile(ctx, hist_bins=4096).text() == g("tile")
    assert mipaf.PafSet.from_text(g("tile")).trim(ctx, "0.2").text() == g("trim")
    assert mipaf.PafSet.from_text(g("trim")).filter(max_tile_level=1).text() == g("primary")
    assert mipaf.PafSet.from_text(g("primary")).chain(ctx).text() == g("rechain")
    assert mipaf.PafSet.from_text(g("input")).chain_tile_trim_filter(ctx, None, "0.2", 10000).text() == g("output")


def test_chain_stage_differential_fuzz_slice():
    """A slice of scripts/gpu_chain_fuzz.py (random PAF sets x random chain / trim parameters, every sub-command and the whole
    job against the oracle); 1 800 cases of the full script ran clean during development."""
    import sys
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gpu_chain_fuzz.py"), "40", "5000"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "40 cases, 0 mismatches" in p.stdout
