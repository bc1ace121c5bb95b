"""CPU tests of the host logic around the hot path: parameter selection by distance, the process
boundary (cactus_call), the local Toil stand-in, chunk / dechunk formats, the PAF validator and the
synthetic generator."""
import os

import numpy as np
import pytest

from cactus_amd import gen, pafcheck
from cactus_amd.paf import chunking
from cactus_amd.paf.local_alignment import select_lastz_params, combine_chunks
from cactus_amd.shared import configWrapper
from cactus_amd.shared.common import cactus_call, getOptionalAttrib
from localjob import FileID, LocalFileStore, LocalJob


def test_distance_selects_the_reference_parameter_sets():
    # local_alignment.py:44-51 with divergences one..five = 0.05..0.25 (cactus_progressive_config.xml:10-13)
    cfg = configWrapper.load_config()
    la = cfg.find("blast").find("lastzArguments").attrib
    ka = cfg.find("blast").find("kegalignArguments").attrib
    for d, key in [(0.0, "one"), (0.05, "one"), (0.0501, "two"), (0.1, "two"), (0.15, "three"), (0.176, "four"), (0.2, "four"),
                   (0.25, "five"), (0.2501, "default"), (3.0, "default")]:
        assert select_lastz_params(d, cfg, 0) == la[key]
        assert select_lastz_params(d, cfg, 2) == ka[key]
    cfg.find("constants").find("divergences").attrib["useDefault"] = "1"
    assert select_lastz_params(0.01, cfg, 0) == la["default"]
    assert "--queryhspbest" not in ka["default"]          # cactus_progressive_config.xml:138


def test_get_optional_attrib():
    cfg = configWrapper.load_config()
    b = cfg.find("blast")
    assert getOptionalAttrib(b, "gpu", typeFn=int, default=5) == 0
    assert getOptionalAttrib(b, "cpu", typeFn=int, default=None) is None
    assert getOptionalAttrib(b, "chunkSize", typeFn=int) == 30000000
    with pytest.raises(RuntimeError):
        getOptionalAttrib(b, "nope", errorIfNotPresent=True)


def test_cactus_call_contract(tmp_path):
    out = tmp_path / "o.txt"
    assert cactus_call(["sh", "-c", "echo hi; echo warn >&2"], outfile=str(out)) is None
    assert out.read_text() == "hi\n"
    assert cactus_call(["sh", "-c", "echo hi; echo warn >&2"], outfile=str(out), returnStdErr=True, outappend=True) == "warn\n"
    assert out.read_text() == "hi\nhi\n"
    assert cactus_call(["sh", "-c", "echo x"], check_output=True) == "x\n"
    with pytest.raises(RuntimeError) as e:
        cactus_call(["sh", "-c", "echo boom >&2; exit 7"])
    assert "exited 7" in str(e.value) and "boom" in str(e.value)


def test_chunk_then_dechunk_round_trip(tmp_path):
    rng = np.random.default_rng(3)
    recs = [("id=E|chrA", gen.random_sequence(2500, rng)), ("id=E|chrB desc", gen.random_sequence(700, rng)),
            ("id=E|chrC", gen.random_sequence(1000, rng))]
    fa = tmp_path / "g.fa"
    gen.write_fasta(str(fa), recs)
    files = chunking.fasta_chunk(str(fa), str(tmp_path / "chunks"), 1000, 100)
    seen = {}
    for f in files:
        for name, seq in pafcheck.read_fasta(f).items():
            base, n, start = name.rsplit("|", 2)
            assert len(seq) <= 1100
            seen.setdefault(base, []).append((int(start), seq, int(n)))
    full = {name.split()[0]: s.tobytes().decode() for name, s in recs}
    for base, parts in seen.items():
        for start, seq, n in parts:
            assert n == len(full[base]) and full[base][start:start + len(seq)] == seq
        assert sorted(s for s, _, _ in parts) == list(range(0, len(full[base]), 1000))
    line = "id=E|chrA|2500|1000\t1100\t10\t60\t-\tid=F|x|9000|3000\t1100\t5\t55\t50\t50\t255\tAS:i:4550\tcg:Z:50=\n"
    f = chunking.paf_dechunk_line(line).split("\t")
    assert f[:9] == ["id=E|chrA", "2500", "1010", "1060", "-", "id=F|x", "9000", "3005", "3055"]
    f = chunking.paf_dechunk_line(line, query_only=True).split("\t")
    assert f[0] == "id=E|chrA" and f[5] == "id=F|x|9000|3000" and f[7] == "5"


def test_combine_chunks_batches_and_dechunks(tmp_path):
    fs = LocalFileStore(str(tmp_path))
    job = LocalJob(fs)
    ids = []
    for k in range(5):
        p = tmp_path / ("c%d.paf" % k)
        p.write_text("q|100|%d\t50\t0\t10\t+\tt|200|0\t200\t0\t10\t10\t10\t255\tAS:i:910\tcg:Z:10=\n" % (k * 10))
        ids.append(fs.writeGlobalFile(str(p)))
    out = combine_chunks(job, ids, 2)          # 5 >= 2*2 -> batched path
    lines = open(str(out)).read().splitlines()
    assert len(lines) == 5 and [l.split("\t")[2] for l in lines] == ["0", "10", "20", "30", "40"]
    assert all(l.split("\t")[0] == "q" and l.split("\t")[1] == "100" for l in lines)


def test_local_job_shim(tmp_path):
    job = LocalJob(LocalFileStore(str(tmp_path)))
    p = tmp_path / "x"
    p.write_text("abc")
    fid = job.fileStore.writeGlobalFile(str(p))
    assert isinstance(fid, FileID) and fid.size == 3
    dst = tmp_path / "y"
    assert job.fileStore.readGlobalFile(fid, str(dst)) == str(dst) and dst.read_text() == "abc"
    assert job.addChildJobFn(lambda j, a: a + 1, 41).rv() == 42


def test_paf_validator_rejects_broken_records():
    good = "q\t100\t0\t20\t+\tt\t100\t5\t25\t20\t20\t255\tAS:i:1820\tcg:Z:20=\n"
    pafcheck.check_paf(good)
    for bad in (good.replace("cg:Z:20=", "cg:Z:19="), good.replace("\t20\t20\t255", "\t19\t20\t255"),
                good.replace("cg:Z:20=", "cg:Z:10=10="), good.replace("cg:Z:20=", "cg:Z:10=0X10=")):
        with pytest.raises(AssertionError):
            pafcheck.check_paf(bad)
    minus = "q\t100\t10\t30\t-\tt\t100\t5\t27\t20\t22\t255\tAS:i:1\tcg:Z:10=2D10=\n"
    pafcheck.check_paf(minus)


def test_generator_is_deterministic_and_has_the_advertised_features():
    a = gen.make_pair(50000, 42)
    b = gen.make_pair(50000, 42)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    t, q = a
    assert abs(np.mean(t > 96) - 0.2) < 0.08 and (t == ord("N")).sum() >= 250 and len(q) != len(t)
    r = gen.make_pair(1000, 1, homologous=False)
    assert len(r[0]) == len(r[1]) == 1000


def test_accelerator_string_is_rocm():
    assert configWrapper.accelerator_string(0) is None and configWrapper.accelerator_string(4) == "rocm:4"


def test_paf_invert_and_validity():
    from cactus_amd.paf.chunking import paf_invert_line
    plus = "q\t100\t10\t35\t+\tt\t200\t5\t27\t15\t27\t255\tAS:i:1\tcg:Z:10=2D5X3I5=2I\n"
    inv = paf_invert_line(plus)
    f = inv.split("\t")
    assert f[:9] == ["t", "200", "5", "27", "+", "q", "100", "10", "35"] and f[-1].strip() == "cg:Z:10=2I5X3D5=2D"
    pafcheck.check_paf(inv)
    minus = "q\t100\t10\t35\t-\tt\t200\t5\t27\t15\t27\t255\tAS:i:1\tcg:Z:10=2D5X3I5=2I\n"
    pafcheck.check_paf(minus)
    inv = paf_invert_line(minus)
    assert inv.split("\t")[-1].strip() == "cg:Z:2D5=3D5X2I10="
    pafcheck.check_paf(inv)
    assert paf_invert_line(paf_invert_line(minus)) == minus


def test_unaligned_bed_and_extract_round_trip(tmp_path):
    rng = np.random.default_rng(5)
    fa = tmp_path / "in.fa"
    recs = [("id=I|c1", gen.random_sequence(3000, rng)), ("id=I|c2", gen.random_sequence(800, rng))]
    gen.write_fasta(str(fa), recs)
    paf = tmp_path / "a.paf"
    paf.write_text("id=I|c1\t3000\t500\t1200\t+\tt\t9\t0\t9\t1\t1\t255\n" "id=I|c1\t3000\t1150\t1300\t-\tt\t9\t0\t9\t1\t1\t255\n"
                   "id=I|c1\t3000\t2950\t3000\t+\tt\t9\t0\t9\t1\t1\t255\n")
    bed = chunking.paf_to_bed_unaligned(str(paf), str(fa), 100)
    assert bed == [("id=I|c1", 0, 500), ("id=I|c1", 1300, 2950), ("id=I|c2", 0, 800)]
    assert chunking.paf_to_bed_unaligned(str(paf), str(fa), 600) == [("id=I|c1", 1300, 2950), ("id=I|c2", 0, 800)]
    sub = tmp_path / "sub.fa"
    chunking.fasta_extract(bed, str(fa), str(sub), 100)
    got = pafcheck.read_fasta(str(sub))
    full = {n: s.tobytes().decode() for n, s in recs}
    assert list(got) == ["id=I|c1|3000|0", "id=I|c1|3000|1200", "id=I|c2|800|0"]
    assert got["id=I|c1|3000|0"] == full["id=I|c1"][0:600] and got["id=I|c1|3000|1200"] == full["id=I|c1"][1200:3000]
    # an alignment on an extracted piece maps back with dechunk --query
    line = "id=I|c1|3000|1200\t1800\t10\t60\t+\tt\t9\t0\t9\t50\t50\t255\tAS:i:1\tcg:Z:50=\n"
    assert chunking.paf_dechunk_line(line, query_only=True).split("\t")[:4] == ["id=I|c1", "3000", "1210", "1260"]


def test_cactus_call_pipes_commands_like_the_reference(tmp_path):
    # local_alignment.py:684-691 passes a list of commands; stdout of the last goes to outfile, any non-zero exit raises
    from cactus_amd.shared.common import cactus_call
    assert cactus_call([["printf", "b\\na\\nc\\n"], ["sort"], ["head", "-n", "2"]], check_output=True) == "a\nb\n"
    out = tmp_path / "o.txt"
    cactus_call([["printf", "x\\n"], ["cat"]], outfile=str(out))
    cactus_call([["printf", "y\\n"]], outfile=str(out), outappend=True)
    assert out.read_text() == "x\ny\n"
    with pytest.raises(RuntimeError) as e:
        cactus_call([["printf", "x\\n"], ["sh", "-c", "cat >/dev/null; echo boom >&2; exit 3"], ["cat"]], check_output=True)
    assert "exited 3" in str(e.value) and "boom" in str(e.value)


def test_inprocess_chaining_job_passes_the_config_values_to_the_c_abi(tmp_path, monkeypatch):
    # chain_tile_trim_filter_one_contig with MIBLAST_INPROCESS=1: one mipaf_chain_tile_trim_filter call carrying the <blast> attributes
    import xml.etree.ElementTree as ET
    from cactus_amd import miblast, mipaf
    from cactus_amd.paf import local_alignment as la
    calls = {}

    class FakeCtx:
        def __init__(self, device):
            calls["device"] = device

        def close(self):
            calls["closed"] = True

    class FakeSet:
        @classmethod
        def from_file(cls, path):
            calls["input"] = open(path).read()
            return cls()

        def chain_tile_trim_filter(self, ctx, cp, identity, min_score, output_secondary=False):
            calls["args"] = (cp.max_gap_length, cp.gap_open, cp.gap_extend, cp.trim_fraction, identity, min_score, output_secondary)

        def write(self, path):
            open(path, "w").write("out\n")

        def close(self):
            pass

    monkeypatch.setattr(miblast, "Context", FakeCtx)
    monkeypatch.setattr(mipaf, "PafSet", FakeSet)
    monkeypatch.setenv("MIBLAST_INPROCESS", "1")
    params = ET.parse(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cactus_amd", "blast_config.xml")).getroot()
    job = LocalJob()
    src = tmp_path / "in.paf"
    src.write_text("x\n")
    fid = job.fileStore.writeGlobalFile(str(src))
    out = la.chain_tile_trim_filter_one_contig(job, fid, "Anc0", params)
    assert open(str(out)).read() == "out\n" and calls["input"] == "x\n" and calls["closed"] and calls["device"] == 0
    assert calls["args"] == (1000000, 5000, 1, 1.0, "0.2", 10000, False)           # cactus_progressive_config.xml:108-113
    assert not os.path.exists(str(fid))


def test_initLastz_resolves_gpu_cores_and_memory_like_the_reference(monkeypatch):
    """ConfigWrapper.initLastz (/root/reference/src/cactus/shared/configWrapper.py:281-393): --gpu N|all -> <blast gpu> and the
    lastzRepeatMask preprocessors; cores default to the whole machine with GPUs on (one process drives them all); --lastzCores /
    --lastzMemory land where make_chunked_alignments reads them; realign goes off; bad values and --latest raise."""
    import copy
    import types
    import xml.etree.ElementTree as ET
    base = configWrapper.load_config()
    ET.SubElement(base, "preprocessor", preprocessJob="lastzRepeatMask", gpu="0", active="0")
    base.find("blast").attrib["realign"] = "1"
    monkeypatch.setattr(configWrapper, "count_amd_gpus", lambda: 8)
    ns = types.SimpleNamespace

    cfg = configWrapper.initLastz(copy.deepcopy(base), ns(gpu="all", batchSystem="single_machine", maxCores=None, lastzCores=None, lastzMemory=None))
    b = cfg.find("blast")
    assert b.attrib["gpu"] == "8" and cfg.find("preprocessor").attrib["gpu"] == "8"
    assert int(b.attrib["cpu"]) == configWrapper.cactus_cpu_count() and b.attrib["realign"] == "0"
    assert "lastz_memory" not in b.attrib

    cfg = configWrapper.initLastz(copy.deepcopy(base), ns(gpu=2, batchSystem="slurm", maxCores=None, lastzCores=16, lastzMemory=12345))
    b = cfg.find("blast")
    assert (b.attrib["gpu"], b.attrib["cpu"], b.attrib["lastz_memory"]) == ("2", "16", "12345")
    assert cfg.find("preprocessor").attrib["cpu"] == "16" and cfg.find("preprocessor").attrib["lastz_memory"] == "12345"

    cfg = configWrapper.initLastz(copy.deepcopy(base), ns(gpu=4, batchSystem="single_machine", maxCores=24, lastzCores=None, lastzMemory=None))
    assert cfg.find("blast").attrib["cpu"] == "24"
    cfg = configWrapper.initLastz(copy.deepcopy(base), ns(gpu=None, batchSystem="single_machine", maxCores=None, lastzCores=None, lastzMemory=None))
    assert cfg.find("blast").attrib["gpu"] == "0" and "cpu" not in cfg.find("blast").attrib and cfg.find("blast").attrib["realign"] == "1"
    legacy = copy.deepcopy(base)
    legacy.find("blast").attrib["gpu"] = "true"                                   # old boolean configs (:322-325)
    assert configWrapper.initLastz(legacy, ns(gpu=None, batchSystem="single_machine")).find("blast").attrib["gpu"] == "8"
    assert configWrapper.initLastz(copy.deepcopy(base), 3).find("blast").attrib["gpu"] == "3"       # bare --gpu value

    for bad, msg in ((ns(gpu="many", batchSystem="single_machine"), "Invalid value"), (ns(gpu=9, batchSystem="single_machine"), "only 8 visible"),
                     (ns(gpu="all", batchSystem="slurm"), "--gpu N required"), (ns(gpu=2, batchSystem="slurm", lastzCores=None), "--lastzCores must be used"),
                     (ns(gpu=1, batchSystem="single_machine", latest=True), "--latest")):
        with pytest.raises(RuntimeError, match=msg):
            configWrapper.initLastz(copy.deepcopy(base), bad)
    monkeypatch.setattr(configWrapper, "count_amd_gpus", lambda: 0)
    with pytest.raises(RuntimeError, match="Unable to automatically determine"):
        configWrapper.initLastz(copy.deepcopy(base), ns(gpu="all", batchSystem="single_machine"))


def test_chunk_scale_workloads_are_the_ones_the_oracle_digests_were_made_from():
    """cactus_amd/workloads.py (chr20 = BASELINE configs[3], hm = the configs[4] stand-in) must generate, on this numpy, the FASTA
    bytes whose chunk pairs the committed oracle digests describe (tests/golden/<key>_pairs.json, scripts/oracle_chunk_digests.py):
    bench.py and the GPU suite compare every chunk pair's PAF with them.  Also: the chunker packs records exactly as
    cactus_amd.paf.chunking.fasta_chunk does."""
    import hashlib
    import json
    from cactus_amd import workloads
    from cactus_amd.paf import chunking
    for key, n_pairs in (("hm", 42), ("chr20", 9)):
        w = workloads.by_name(key)
        gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"{key}_pairs.json")))
        assert len(w.pairs) == n_pairs == len(gold["pairs"]) and all(p is not None for p in gold["pairs"])
        assert gold["fasta_md5"] == hashlib.md5(b"".join(w.tfa + w.qfa)).hexdigest(), key
        assert gold["options"] == w.options
    # the in-memory chunker against the file-based one (faffy chunk's packing rule)
    import tempfile
    recs = [("a", np.frombuffer(b"ACGT" * 700, dtype=np.uint8)), ("b", np.frombuffer(b"TTGCA" * 90, dtype=np.uint8)), ("c", np.frombuffer(b"G" * 1300, dtype=np.uint8))]
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "g.fa")
        open(src, "wb").write(gen.fasta_bytes(recs))
        files = chunking.fasta_chunk(src, os.path.join(d, "chunks"), 1000, 100)
        names_files = [[l[1:].strip() for l in open(f) if l.startswith(">")] for f in files]
    assert names_files == [[n for n, _ in f] for f in workloads.chunk_records(recs, 1000, 100)]


def test_env_table_is_the_one_the_code_gives():
    """ENV.md ("one documented table generated from the code") is scripts/env_table.py's output for the sources
    as they are: a switch added or moved without regenerating the table fails here."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "env_table.py")], capture_output=True, text=True, cwd=root)
    assert out.returncode == 0, out.stderr
    assert out.stdout == open(os.path.join(root, "ENV.md")).read(), "run: python scripts/env_table.py > ENV.md"


def test_container_modes_hand_the_gpus_over_the_amd_way(tmp_path, monkeypatch):
    """common.py:536-560 (singularity exec ... --nv) and :646-705 (docker run ... --gpus N | --gpus "device=$SLURM_JOB_GPUS"): the same
    call shapes with the AMD way of giving a container its GPUs -- the driver's device nodes + groups and ROCR_VISIBLE_DEVICES for docker,
    --rocm for singularity (last row) -- and nothing GPU-related on a CPU job."""
    from cactus_amd.shared import common
    argv = ['run_kegalign', 'A.fa', 'B.fa', '--format=paf:wfmash', '--num_gpu', '2', '--num_threads', '8']
    d = common.dockerCommand(tool='quay.io/cactus:amd', work_dir=str(tmp_path), parameters=argv, gpus=2, cpus=8, environ={})
    assert d[:2] == ['docker', 'run'] and d[-len(argv):] == argv and d[-len(argv) - 1] == 'quay.io/cactus:amd'
    assert '--device=/dev/kfd' in d and '--device=/dev/dri' in d and '--gpus' not in d
    assert d[d.index('--group-add') + 1] == 'video' and 'render' in d
    assert 'ROCR_VISIBLE_DEVICES=0,1' in d and d[d.index('--cpus') + 1] == '8'
    assert d[d.index('-v') + 1] == '{}:/data'.format(tmp_path) and d[d.index('--entrypoint') + 1] == '/opt/cactus/wrapper.sh' and '--rm' in d
    # Slurm names the GPUs of the job (the reference: --gpus "device=$SLURM_JOB_GPUS")
    assert 'ROCR_VISIBLE_DEVICES=4,5' in common.dockerCommand(tool='img', work_dir=str(tmp_path), parameters=argv, gpus=2, environ={'SLURM_JOB_GPUS': '4,5'})
    cpu_job = common.dockerCommand(tool='img', work_dir=str(tmp_path), parameters=['lastz', 'a', 'b'], gpus=0, environ={})
    assert not any('kfd' in a or 'ROCR' in a for a in cpu_job)
    s = common.singularityCommand(tool='/img/cactus.sif', work_dir=str(tmp_path), parameters=argv, gpus=2)
    assert s[:3] == ['singularity', '--silent', 'exec'] and '--rocm' in s and '--nv' not in s and s[-len(argv):] == argv
    assert s[s.index('-B') + 1] == '{}:/mnt'.format(tmp_path) and s[s.index('--pwd') + 1] == '/mnt'
    assert '--rocm' not in common.singularityCommand(tool='/img/cactus.sif', work_dir=str(tmp_path), parameters=['lastz'], gpus=0)
    # cactus_call wraps the command under CACTUS_BINARIES_MODE=docker|singularity: a stand-in `docker` on PATH records what it is given
    fake = tmp_path / "bin"
    fake.mkdir()
    for name in ("docker", "singularity"):
        (fake / name).write_text("#!/bin/sh\necho \"$0 $@\" > {}/{}.argv\n".format(tmp_path, name))
        (fake / name).chmod(0o755)
    env = dict(os.environ, CACTUS_BINARIES_MODE="docker", CACTUS_DOCKER_IMAGE="quay.io/cactus:amd", PATH=str(fake) + os.pathsep + os.environ["PATH"])
    monkeypatch.setattr(common, "BIN_DIR", str(fake))
    common.cactus_call(parameters=argv, work_dir=str(tmp_path), gpus=2, env=env)
    seen = (tmp_path / "docker.argv").read_text()
    assert '--device=/dev/kfd' in seen and 'quay.io/cactus:amd run_kegalign A.fa B.fa' in seen
    env.update(CACTUS_BINARIES_MODE="singularity", CACTUS_SINGULARITY_IMG="/img/cactus.sif")
    common.cactus_call(parameters=argv, work_dir=str(tmp_path), gpus=2, env=env)
    assert '--rocm /img/cactus.sif run_kegalign' in (tmp_path / "singularity.argv").read_text()
    # prepareWorkDir (common.py:695-730): absolute paths under the work directory are rewritten relative to the mount, also inside one argument
    # that holds several; without a work_dir it is derived from the arguments that exist
    (tmp_path / "A.fa").write_text(">a\nACGT\n"); (tmp_path / "sub").mkdir(); (tmp_path / "sub" / "B.fa").write_text(">b\nACGT\n")
    wd, pars = common.container_work_dir(str(tmp_path), ['lastz', str(tmp_path / "A.fa") + '[multiple]', '%s %s' % (tmp_path / "A.fa", tmp_path / "sub" / "B.fa"), '--x'])
    assert wd == str(tmp_path) and pars == ['lastz', 'A.fa[multiple]', 'A.fa sub/B.fa', '--x']
    wd, pars = common.container_work_dir(None, ['paffy', 'chain', '-i', str(tmp_path / "A.fa")])
    assert wd == str(tmp_path) and pars == ['paffy', 'chain', '-i', 'A.fa']
    wd, pars = common.container_work_dir(None, ['faffy', str(tmp_path / "A.fa"), str(tmp_path / "sub" / "B.fa")])
    assert wd.rstrip('/') == str(tmp_path) and pars[1:] == ['A.fa', 'sub/B.fa']
    assert common.container_work_dir(None, ['lastz', '--help'])[0] == os.getcwd()
    # a piped command list is ONE container running bash -c 'set -eo pipefail && a | b' (common.py:764-778), paths relative to the mount
    env.update(CACTUS_BINARIES_MODE="docker")
    common.cactus_call(parameters=[['paffy', 'invert', '-i', str(tmp_path / "A.fa")], ['paffy', 'chain', '--maxGapLength', '10']], work_dir=str(tmp_path), env=env)
    seen = (tmp_path / "docker.argv").read_text()
    assert seen.count('docker run') == 1 and '--entrypoint /bin/bash' in seen and seen.rstrip().endswith("quay.io/cactus:amd -c set -eo pipefail && paffy invert -i A.fa | paffy chain --maxGapLength 10")
    env.update(CACTUS_BINARIES_MODE="singularity")
    common.cactus_call(parameters=[['paffy', 'invert', '-i', str(tmp_path / "A.fa")], ['paffy', 'chain']], work_dir=str(tmp_path), env=env)
    assert (tmp_path / "singularity.argv").read_text().rstrip().endswith("/img/cactus.sif bash -c set -eo pipefail && paffy invert -i A.fa | paffy chain")
    env.pop("CACTUS_SINGULARITY_IMG")
    with pytest.raises(RuntimeError):
        common.cactus_call(parameters=argv, work_dir=str(tmp_path), gpus=2, env=env)


def test_bench_line_is_compact_and_carries_the_contract():
    """SURVEY 8d / round-5 review: the driver reads an 8 KB tail of stdout, and round 5's 20 KB line did not parse.  bench.compact_line() of a
    full result object (round 5's own, every leg present) must be ONE line below 8 000 bytes with the contract's keys, `roofline` and
    `cpu_baseline`; the full object goes to the file the line names."""
    import importlib.util
    import json
    import io
    import contextlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for name in ("r05_bench.json", "r04_bench.json", "r05_hm_bench_under_rocprof.json"):
        full = json.loads(open(os.path.join(root, "profiles", name)).read().strip().splitlines()[-1])
        assert len(json.dumps(full)) > 8000 or "hm" in name
        text = bench.compact_line(full, "bench_full.json")
        assert "\n" not in text and len(text.encode()) < 8000
        line = json.loads(text)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
            assert k in line, k
        assert line["unit"] == "Gcell/s" and line["dtype"] == "int32" and line["vs_baseline"] is None and "workload" in line["config"]
        assert "OUTSIDE the step" in line["config"]["timed_region"]
        r = line["roofline"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launch_ms", "algorithmic_bytes_per_launch"):
            assert k in r, k
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 * r["frac"] and "frac_of_measured_peak" in r["valu"]
        assert abs(line["value"] - full["value"]) < 1e-5 * full["value"] and line["full"] == "bench_full.json"
        if "cpu_baseline" in full:
            cb = line["cpu_baseline"]
            assert cb["kind"] == "port" and cb["cores"] == full["cpu_baseline"]["cores"] and cb["same_bytes"] is True and "sample" in cb and cb["value"] > 0
        if "hm" in full:
            for leg in ("chr20", "hm"):
                assert line["legs"][leg]["parity"]["same_bytes"] is True and line["legs"][leg]["ms_per_step"] > 0 and "hbm_read_frac" in line["legs"][leg]
    # a grotesquely long workload description cannot push the line over the limit either
    full["config"]["workload"] = "x" * 50000
    assert len(bench.compact_line(full, "f").encode()) < 8000
    # emit(): the file holds the full object, stdout exactly the one line
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            bench.emit(full, os.path.join(d, "full.json"))
        assert buf.getvalue().count("\n") == 1 and json.loads(buf.getvalue())["full"] == os.path.join(d, "full.json")
        assert json.load(open(os.path.join(d, "full.json"))) == full
