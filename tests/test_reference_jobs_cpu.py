"""The boundary tested through the REFERENCE's own statements: the job functions of /root/reference/src/cactus/paf/local_alignment.py,
imported unmodified (tests/refjobs.py supplies stand-ins for toil / sonLib / Bio and a local-binaries cactus_call), run against the
front ends of <repo>/bin found on PATH, as CACTUS_BINARIES_MODE=local finds them.  There is no GPU in this container, so `lastz` /
`run_kegalign` on PATH are a two-line shim: bin/lastz first, with MIBLAST_PARSE_ONLY=1 (our real argv parser must accept the
reference's command line and find both files), then the CPU oracle's front end for the alignment itself.  Everything else -- faffy
chunk, paffy dechunk / invert / to_bed / upconvert, faffy extract -- is the product's own host code.  The same argv lists are committed
(tests/golden/ref_argv.json, written by tests/golden/make_ref_argv.py) and replayed against the real bin/lastz on the MI355X by
tests/test_parity_gpu.py.  Skipped where /root/reference is absent (the GPU box)."""
import os
import stat
import sys
import xml.etree.ElementTree as ET

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refjobs  # noqa: E402
from localjob import LocalJob  # noqa: E402
from cactus_amd import gen  # noqa: E402

pytestmark = pytest.mark.skipif(not refjobs.available(), reason="/root/reference is not here (the reference's job functions cannot be imported)")
CONFIG = "/root/reference/src/cactus/cactus_progressive_config.xml"


@pytest.fixture(scope="module")
def ref(tmp_path_factory):
    import subprocess
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s"], check=True)
    shim = tmp_path_factory.mktemp("shim")
    for name in ("lastz", "run_kegalign"):
        p = shim / name
        # (the oracle's front end has no --num_gpu / --num_threads: the shim drops them after bin/<name> has accepted them)
        p.write_text("#!/bin/bash\nMIBLAST_PARSE_ONLY=1 %s/bin/%s \"$@\" >/dev/null || exit $?\nargs=()\nwhile [ $# -gt 0 ]; do case \"$1\" in --num_gpu|--num_threads) shift 2;; *) args+=(\"$1\"); shift;; esac; done\n"
                     "exec %s/oracle/oracle_lastz \"${args[@]}\"\n" % (ROOT, name, ROOT))
        p.chmod(p.stat().st_mode | stat.S_IEXEC)
    refjobs.path_dirs[:] = [str(shim), os.path.join(ROOT, "bin")]
    refjobs.calls.clear()
    return refjobs.load()


def params(**blast_over):
    cfg = ET.parse(CONFIG).getroot()
    for k, v in blast_over.items():
        cfg.find("blast").attrib[k] = str(v)
    return cfg


def genome_files(job, tmp_path, n=60000, seed=5, contigs=2):
    out = []
    for g, (name, sd) in enumerate((("A", seed), ("B", seed + 1))):
        t, q = gen.make_pair(n, seed)
        seq = t if g == 0 else q
        recs = [(f"id={name}|chr{k}", seq[k * (len(seq) // contigs):(k + 1) * (len(seq) // contigs)]) for k in range(contigs)]
        p = tmp_path / f"{name}.fa"
        p.write_bytes(gen.fasta_bytes(recs))
        out.append(job.fileStore.writeGlobalFile(str(p)))
    return out


@pytest.mark.parametrize("distance,gpu", [(0.03, 0), (0.12, 0), (0.2, 0), (0.6, 0), (0.6, 1)])
def test_reference_run_lastz_drives_our_front_end(ref, olz, tmp_path, distance, gpu):
    from cactus_amd import miblast
    from cactus_amd.paf.local_alignment import select_lastz_params
    from cactus_amd.shared.configWrapper import load_config
    job = LocalJob()
    a, b = genome_files(job, tmp_path, 30000, 7, 1)
    cfg = params(gpu=gpu) if gpu else params()
    refjobs.calls.clear()
    out = ref.run_lastz(job, "A", a, "B", b, distance, cfg)
    (cmds, _), = refjobs.calls
    argv = cmds[0]
    assert argv[0] == ("run_kegalign" if gpu else "lastz") and argv[3] == "--format=paf:wfmash"
    assert argv[1] == ("A.fa" if gpu else "A.fa[multiple][nameparse=darkspace]")
    # the option string the reference picked is the one our mirror of the selection picks from OUR copy of the option sets
    ours = select_lastz_params(distance, load_config(), gpu)
    assert argv[4:] == ours.split(" ") + (["--num_gpu", str(gpu), "--num_threads", "1"] if gpu else [])
    pm = miblast.params_from_args([x for x in argv[4:] if x])
    want = olz.align(open(str(a), "rb").read(), open(str(b), "rb").read(), olz.default_params(**{f: getattr(pm, f) for f, _ in pm._fields_}), details=False)["paf"]
    assert open(str(out), "rb").read() == want


def test_reference_make_chunked_alignments_and_combine_chunks(ref, olz, tmp_path):
    """faffy chunk (bin/faffy), one run_lastz per chunk pair, paffy dechunk per chunk file (bin/paffy): the reference's functions, our tools"""
    from cactus_amd import miblast
    from cactus_amd.paf import chunking
    job = LocalJob()
    a, b = genome_files(job, tmp_path, 50000, 9, 2)
    cfg = params(chunkSize=12000, overlapSize=500)
    refjobs.calls.clear()
    out = ref.make_chunked_alignments(job, "A", a, "B", b, 0.6, cfg)
    tools = [c[0][0][0] + " " + c[0][0][1] if c[0][0][0] != "lastz" else "lastz" for c in refjobs.calls]
    assert tools.count("faffy chunk") == 2 and tools.count("lastz") >= 16 and tools.count("paffy dechunk") == tools.count("lastz")
    got = sorted(open(str(out)).read().splitlines(True))
    # what it must be: the oracle on every (chunk of A, chunk of B) pair, dechunked -- chunks made by the Python core of the chunker
    ca = chunking.fasta_chunk(str(a), str(tmp_path / "ca"), 12000, 500)
    cb = chunking.fasta_chunk(str(b), str(tmp_path / "cb"), 12000, 500)
    pm = miblast.params_from_args(cfg.find("blast").find("lastzArguments").attrib["default"].split())
    po = olz.default_params(**{f: getattr(pm, f) for f, _ in pm._fields_})
    want = []
    for fa in ca:
        for fb in cb:
            paf = olz.align(open(fa, "rb").read(), open(fb, "rb").read(), po, details=False)["paf"].decode()
            want += [chunking.paf_dechunk_line(l) for l in paf.splitlines(True)]
    assert got == sorted(want) and len(got) > 5


class Event:
    def __init__(self, iD):
        self.iD = iD


def test_reference_outgroup_chain_equals_the_mirror(ref, tmp_path):
    """make_ingroup_to_outgroup_alignments_0..3 of the reference (chunked alignment to the first outgroup, `paffy to_bed --excludeAligned |
    faffy extract`, the rest against the next outgroup, `paffy dechunk --query`, `paffy invert`) against the same chain through
    cactus_amd.paf.local_alignment's functions: same bytes."""
    from cactus_amd.paf import local_alignment as mirror
    job = LocalJob()
    files = {}
    for name, seed in (("ing", 21), ("out1", 21), ("out2", 22)):
        t, q = gen.make_pair(40000, seed, sub_rate=0.08 if name != "out2" else 0.15)
        p = tmp_path / f"{name}.fa"
        p.write_bytes(gen.fasta_bytes([(f"id={name}|chr0", t if name == "ing" else q)]))
        files[name] = job.fileStore.writeGlobalFile(str(p))
    cfg = params(chunkSize=15000, overlapSize=500)
    ing, o1, o2 = Event("ing"), Event("out1"), Event("out2")
    dist = {(ing, o1): 0.1, (ing, o2): 0.3}
    refjobs.calls.clear()
    got = ref.make_ingroup_to_outgroup_alignments_0(job, ing, [o1, o2], dict(files), dist, cfg)
    used = {" ".join(c[0][0][:2]) for c in refjobs.calls}
    assert {"paffy to_bed", "faffy extract", "paffy dechunk", "paffy invert", "faffy chunk"} <= used
    os.environ["PATH"] = os.pathsep.join(refjobs.path_dirs) + os.pathsep + os.environ["PATH"]      # (the mirror's cactus_call puts bin/ first; the shim must still win for lastz)
    import cactus_amd.shared.common as common
    saved = common.BIN_DIR
    common.BIN_DIR = os.pathsep.join(refjobs.path_dirs)
    try:
        want = mirror.make_ingroup_to_outgroup_alignments_0(job, "ing", ["out1", "out2"], dict(files), {("ing", "out1"): 0.1, ("ing", "out2"): 0.3}, cfg)
    finally:
        common.BIN_DIR = saved
    a, b = sorted(open(str(got)).read().splitlines()), sorted(open(str(want)).read().splitlines())
    assert a == b and len(a) > 3


def test_reference_trim_unaligned_sequences(ref, tmp_path):
    """trim_unaligned_sequences of the reference over bin/paffy to_bed, bin/faffy extract, bin/paffy upconvert == the per-base oracle"""
    import subprocess
    from test_text_oracle_cpu import two_genome_case
    files, paf = two_genome_case(3)
    job = LocalJob()
    ids, paths = [], []
    for k, recs in enumerate(files):
        p = tmp_path / f"g{k}.fa"
        p.write_bytes(gen.fasta_bytes(recs))
        paths.append(str(p))
        ids.append(job.fileStore.writeGlobalFile(str(p)))
    (tmp_path / "a.paf").write_text(paf)
    cfg = params(trimOutgroupFlanking=50)
    seqs, out = ref.trim_unaligned_sequences(job, ids, job.fileStore.writeGlobalFile(str(tmp_path / "a.paf")), cfg, has_resources=True)
    want = subprocess.run([os.path.join(ROOT, "oracle", "oracle_paffy_text"), "trim_aligned", str(tmp_path / "a.paf"), "50", *paths], capture_output=True, check=True).stdout.decode()
    sections = want.split("== ")
    assert [open(str(s)).read() for s in seqs] == [s.split("\n", 1)[1] for s in sections if s.startswith("file ")]
    assert open(str(out)).read() == [s.split("\n", 1)[1] for s in sections if s.startswith("paf")][0]


def test_mirror_trim_unaligned_sequences_equals_the_reference_job(ref, tmp_path):
    """cactus_amd.paf.local_alignment.trim_unaligned_sequences (the job a deployment without /root/reference runs) gives the files the
    reference's own body gives over the same front ends, and without resources of its own it re-issues itself as a child job with the
    reference's disk / memory request (local_alignment.py:865-869)."""
    from cactus_amd.paf import local_alignment as mirror
    from test_text_oracle_cpu import two_genome_case
    files, paf = two_genome_case(5)
    cfg = params(trimOutgroupFlanking=30)
    results = []
    for fn in (ref.trim_unaligned_sequences, mirror.trim_unaligned_sequences):
        job = LocalJob()
        ids = []
        for k, recs in enumerate(files):
            p = tmp_path / f"h{k}.fa"
            p.write_bytes(gen.fasta_bytes(recs))
            ids.append(job.fileStore.writeGlobalFile(str(p)))
        (tmp_path / "b.paf").write_text(paf)
        seqs, out = fn(job, ids, job.fileStore.writeGlobalFile(str(tmp_path / "b.paf")), cfg, has_resources=fn is ref.trim_unaligned_sequences)
        results.append(([open(str(s)).read() for s in seqs], open(str(out)).read()))
        if fn is mirror.trim_unaligned_sequences:                      # (went through the child job: LocalJob records what was asked for)
            asked = job.children[-1]
            size = sum(i.size for i in ids)
            assert asked["fn"] is mirror.trim_unaligned_sequences and asked["disk"] == 4 * size + 2 * os.path.getsize(str(tmp_path / "b.paf"))
            assert asked["memory"] == max(2 ** 28, asked["disk"])
    assert results[0] == results[1] and len(results[0][0]) == len(files) and results[0][1].count("\n") > 3
