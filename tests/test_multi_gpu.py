"""-m gpu : one job over several devices and over blocks of the inputs (cactus_amd/csrc/mb_multi.cpp, include/miblast.h
miblast_multi) -- the inside of `run_kegalign A.fa B.fa ... --num_gpu G` (/root/reference/src/cactus/paf/local_alignment.py:54-58,
393-405; bigChunkSize, cactus_progressive_config.xml:91).  The contract under test: the PAF bytes equal what ONE oracle run over
the whole files writes, whatever the number of devices, the block size or the dealing of block pairs.  A one-GPU box plays
several devices through $MIBLAST_DEVICE_MAP (logical -> physical ordinals)."""
import os
import subprocess

import numpy as np
import pytest

from cases import KEG_DEFAULT, DEFAULT, multi_contig

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEG_FOUR = "--step=3 --ambiguous=iupac,100,100 --ydrop=3500 --hspthresh=2600 --gappedthresh=2800".split()


def genome_like(seed, n_t=7, n_q=6):
    """Two assemblies of several contigs: every query contig is stitched from mutated pieces of TWO target contigs (one of them
    reverse-complemented now and then), so that a query sequence has alignments in several target blocks and the merge of the
    blocks' lists on anchor order is exercised; plus ragged and empty records."""
    from cactus_amd import gen
    rng = np.random.default_rng(seed)
    tl = [int(x) for x in rng.integers(2500, 9000, size=n_t)]
    trecs = [("id=T|c%d" % i, gen.random_sequence(n, rng)) for i, n in enumerate(tl)]
    trecs.insert(3, ("id=T|tiny", gen.random_sequence(12, rng)))
    qrecs = []
    for j in range(n_q):
        a, b = int(rng.integers(0, n_t)), int(rng.integers(0, n_t))
        pa = gen.mutate(trecs[a if a < 3 else a + 1][1], rng, 0.07, 0.004)
        pb = gen.mutate(trecs[b if b < 3 else b + 1][1], rng, 0.10, 0.006)
        if j % 3 == 1:
            pb = gen.revcomp(pb)
        qrecs.append(("id=Q|s%d" % j, np.concatenate([pa[: len(pa) * 2 // 3], gen.random_sequence(300, rng), pb[len(pb) // 4:]])))
    qrecs.insert(2, ("id=Q|empty", np.zeros(0, dtype=np.uint8)))
    return gen.fasta_bytes(trecs), gen.fasta_bytes(qrecs)


def _oracle(olz, tf, qf, args):
    from cactus_amd import miblast
    pm = miblast.params_from_args(args)
    return pm, olz.align(tf, qf, olz.default_params(**{f: getattr(pm, f) for f, _ in pm._fields_}))


@pytest.mark.parametrize("args,block", [(KEG_DEFAULT, "9000"), (KEG_FOUR, "9000"), (KEG_DEFAULT, "20000"), (KEG_FOUR, None)],
                         ids=["default_9k", "step3_phase_9k", "default_20k", "step3_unblocked"])
def test_blocked_job_equals_one_oracle_run_over_the_whole_files(olz, monkeypatch, args, block):
    """Target and query cut into blocks of whole contigs (forced small here; 2^30 bases in production): bytes and counters of
    the assembled job equal the oracle's single run, including the --step phase of blocks that do not start at a multiple of
    the step and the order of a query's alignments across target blocks."""
    from cactus_amd import miblast
    tf, qf = genome_like(101)
    pm, want = _oracle(olz, tf, qf, args)
    assert want["paf"].count(b"\n") >= 8
    if block:
        monkeypatch.setenv("MIBLAST_BLOCK_BASES", block)
    m = miblast.Multi(1)
    try:
        paf, st = m.align_fasta_pairs([(tf, qf)], pm)
    finally:
        m.close()
    assert paf == want["paf"]
    for k in ("seed_hits", "hits_extended", "ungapped_cols", "hsps", "anchors", "anchors_skipped", "dp_sides", "dp_cells", "dp_rows", "alignments"):
        assert st[k] == want["counters"][k], k


@pytest.mark.parametrize("ngpu", [2, 3])
def test_num_gpu_does_not_change_a_byte(olz, monkeypatch, ngpu):
    """SURVEY 8e determinism: n_gpu in {1, 2, 3} (logical devices sharing this box's GPU) give the oracle's bytes, with and
    without forced target blocks; every device gets work."""
    from cactus_amd import miblast
    tf, qf = genome_like(202, n_t=9, n_q=8)
    pm, want = _oracle(olz, tf, qf, KEG_DEFAULT)
    monkeypatch.setenv("MIBLAST_DEVICE_MAP", ",".join(["0"] * ngpu))
    assert miblast.device_count() == ngpu
    for block in (None, "12000"):
        if block:
            monkeypatch.setenv("MIBLAST_BLOCK_BASES", block)
        m = miblast.Multi(ngpu)
        try:
            paf, st = m.align_fasta_pairs([(tf, qf)], pm)
        finally:
            m.close()
        assert paf == want["paf"], block
        assert st["dp_cells"] == want["counters"]["dp_cells"] and st["seed_hits"] == want["counters"]["seed_hits"]


def test_chunk_pair_list_sharded_over_devices(olz, monkeypatch):
    """miblast_multi_align_fasta_pairs = SURVEY 8b's multi-GPU entry: a list of chunk pairs dealt to the devices, output in pair
    order = the concatenation of the single jobs, for 1 and 2 devices (queryhspbest allowed: single-block targets)."""
    from cactus_amd import miblast
    from cases import pair
    pairs = [pair(20000, 61), multi_contig(62), pair(30000, 63, sub_rate=0.05, indel_rate=0.003), pair(8000, 64, homologous=False), genome_like(65)]
    pm = miblast.params_from_args(DEFAULT)
    want = b"".join(olz.align(tf, qf, olz.default_params(**{f: getattr(pm, f) for f, _ in pm._fields_}))["paf"] for tf, qf in pairs)
    for ngpu in (1, 2):
        monkeypatch.setenv("MIBLAST_DEVICE_MAP", ",".join(["0"] * ngpu))
        m = miblast.Multi(ngpu)
        try:
            paf, _ = m.align_fasta_pairs(pairs, pm)
        finally:
            m.close()
        assert paf == want, ngpu


def test_run_kegalign_front_end_uses_every_gpu_of_the_job(olz, tmp_path, monkeypatch):
    """bin/run_kegalign with the argv of local_alignment.py:54-58 and --num_gpu 2: same bytes as --num_gpu 1 and as the oracle,
    empty stderr (the GPU branch greps it, :75-83)."""
    tf, qf = genome_like(303)
    (tmp_path / "A.fa").write_bytes(tf)
    (tmp_path / "B.fa").write_bytes(qf)
    _, want = _oracle(olz, tf, qf, KEG_DEFAULT)
    env = dict(os.environ, MIBLAST_DEVICE_MAP="0,0", MIBLAST_BLOCK_BASES="15000")
    outs = []
    for ngpu in ("1", "2"):
        p = subprocess.run([os.path.join(ROOT, "bin", "run_kegalign"), str(tmp_path / "A.fa"), str(tmp_path / "B.fa"), "--format=paf:wfmash",
                            *KEG_DEFAULT, "--num_gpu", ngpu, "--num_threads", "2"], capture_output=True, env=env)
        assert p.returncode == 0 and p.stderr == b"", p.stderr
        outs.append(p.stdout)
    assert outs[0] == outs[1] == want["paf"]
    p = subprocess.run([os.path.join(ROOT, "bin", "run_kegalign"), str(tmp_path / "A.fa"), str(tmp_path / "B.fa"), "--format=paf:wfmash",
                        *KEG_DEFAULT, "--num_gpu", "3"], capture_output=True, env=env)
    assert p.returncode != 0 and b"num_gpu" in p.stderr


@pytest.mark.parametrize("limit", [0, 7, 2])
def test_repeat_mask_call_assembled_from_blocks_equals_one_oracle_run(olz, monkeypatch, limit):
    """The repeat masker's call (cactus_lastzRepeatMask.py:97-105: fragments against the assembly, --ungapped --format=general:... --markend,
    --queryhsplimit=keep,nowarn:N) on a target that needs SEVERAL blocks (round 6; refused until then): one header line, the HSPs of every
    query sequence in the order one search over the whole target finds them (query position, word variant, target position descending --
    copies of a repeat in different blocks hit the same query position), at most N per sequence and strand, one end marker: the bytes of ONE
    oracle run over the whole files.  Several query blocks too (two logical devices)."""
    from cactus_amd import gen, miblast
    from cactus_amd.preprocessor.lastz_repeat_mask import fasta_fragments
    rng = np.random.default_rng(31 + limit)
    unit = gen.random_sequence(500, rng)
    trecs = []
    for c in range(6):
        parts = []
        for k in range(5):
            parts.append(gen.random_sequence(int(rng.integers(400, 1500)), rng))
            parts.append(gen.mutate(unit, rng, 0.04, 0.0) if (k + c) % 2 == 0 else gen.revcomp(gen.mutate(unit, rng, 0.05, 0.0)))
        trecs.append(("id=E|c%d" % c, np.concatenate(parts)))
    tf = gen.fasta_bytes(trecs)
    qf = fasta_fragments(tf.decode(), 200, 100, "zero").encode()
    args = "--step=3 --ambiguous=iupac,100,100 --ungapped --format=general:name1,zstart1,end1,name2,zstart2+,end2+ --markend".split()
    if limit:
        args.insert(4, "--queryhsplimit=keep,nowarn:%d" % limit)
    pm, want = _oracle(olz, tf, qf, args)
    assert want["paf"].count(b"\n") > 300 and want["paf"].endswith(b"# lastz end-of-file\n")
    m = miblast.Multi(1)
    try:
        whole, _ = m.align_fasta_pairs([(tf, qf)], pm)
        assert whole == want["paf"]
        monkeypatch.setenv("MIBLAST_BLOCK_BASES", "9000")          # the target in three or four blocks, the fragments in several
        blocked, _ = m.align_fasta_pairs([(tf, qf)], pm)
    finally:
        m.close()
    assert blocked == want["paf"]
    monkeypatch.setenv("MIBLAST_DEVICE_MAP", "0,0")
    m = miblast.Multi(2)
    try:
        two, _ = m.align_fasta_pairs([(tf, qf)], pm)
    finally:
        m.close()
    assert two == want["paf"]


def _copies_of_a_unit(seed):
    """A target whose contigs hold copies of one 600-base unit -- three of them the unit itself and nothing else (HSPs of EQUAL score in
    different target blocks: the extension ends at the contig's ends), others mutated, one reverse-complemented -- and query sequences that
    are the unit, its reverse complement, a mutated copy and a copy inside random sequence: every query sequence has more HSPs than a small
    --queryhspbest keeps, spread over the blocks."""
    from cactus_amd import gen
    rng = np.random.default_rng(seed)
    unit = gen.random_sequence(600, rng)
    rnd = lambda n: gen.random_sequence(n, rng)                      # noqa: E731
    trecs = [("id=T|a", np.concatenate([rnd(1200), gen.mutate(unit, rng, 0.04, 0.0), rnd(900)])),
             ("id=T|u1", unit.copy()),
             ("id=T|b", rnd(2500)),
             ("id=T|m", gen.mutate(unit, rng, 0.03, 0.002)),
             ("id=T|u2", unit.copy()),
             ("id=T|c", np.concatenate([rnd(700), gen.revcomp(unit), rnd(1500), gen.mutate(unit, rng, 0.08, 0.0)])),
             ("id=T|u3", unit.copy()),
             ("id=T|d", np.concatenate([rnd(1000), gen.mutate(unit, rng, 0.06, 0.004), rnd(300)]))]
    qrecs = [("id=Q|u", unit.copy()), ("id=Q|rc", gen.revcomp(unit)), ("id=Q|m", gen.mutate(unit, rng, 0.02, 0.0)),
             ("id=Q|in", np.concatenate([rnd(500), unit, rnd(500)]))]
    return gen.fasta_bytes(trecs), gen.fasta_bytes(qrecs)


@pytest.mark.parametrize("best,ties,limit", [(2, "earlier", 0), (2, "later", 0), (1, "earlier", 0), (5, "earlier", 0), (3, "later", 0),
                                             (0, "earlier", 3), (0, "earlier", 1), (1, "later", 2), (2, "later", 3), (3, "earlier", 4)])
def test_queryhspbest_over_target_blocks_equals_one_oracle_run(olz, monkeypatch, best, ties, limit):
    """--queryhspbest=N (every option set of cactus_progressive_config.xml:131-136 passes it) on a target that needs SEVERAL blocks (round 6;
    refused until then): the N best HSPs of a query sequence and strand are the N best over the WHOLE target -- ranked over the blocks, of
    equal scores the earlier (or, A.9 #11, the later) found in the order one search over the whole target finds them -- and the gapped stage
    starts from exactly those: the bytes and counters of ONE oracle run over the whole files, with one and with two logical devices.
    limit: --queryhsplimit=keep,nowarn:N in front of the gapped stage (and of --queryhspbest) the same way -- the whole target's first N in
    found order."""
    from cactus_amd import miblast
    tf, qf = _copies_of_a_unit(500 + best + 10 * limit)
    args = [a for a in DEFAULT if not a.startswith("--queryhspbest")] + ["--miblast-hspbest-ties=" + ties]
    if best:
        args.append("--queryhspbest=%d" % best)
    if limit:
        args.append("--queryhsplimit=keep,nowarn:%d" % limit)
    pm, want = _oracle(olz, tf, qf, args)
    _, unlimited = _oracle(olz, tf, qf, [a for a in args if not a.startswith("--queryhsp")])
    if best and limit:
        _, only_best = _oracle(olz, tf, qf, [a for a in args if not a.startswith("--queryhsplimit")])
        assert only_best["paf"] != want["paf"]                       # (the limit in front changes what the ranking sees)
    assert want["counters"]["hsps"] < unlimited["counters"]["hsps"] and want["paf"] != unlimited["paf"]          # (the limit binds)
    assert want["paf"].count(b"\n") >= 4
    m = miblast.Multi(1)
    try:
        whole, _ = m.align_fasta_pairs([(tf, qf)], pm)
        assert whole == want["paf"]
        monkeypatch.setenv("MIBLAST_BLOCK_BASES", "3600")          # the unit's exact copies in three different target blocks
        blocked, st = m.align_fasta_pairs([(tf, qf)], pm)
    finally:
        m.close()
    assert blocked == want["paf"]
    for k in ("hsps", "anchors", "anchors_skipped", "dp_sides", "dp_cells", "dp_rows", "alignments"):
        assert st[k] == want["counters"][k], k
    monkeypatch.setenv("MIBLAST_DEVICE_MAP", "0,0")
    m = miblast.Multi(2)
    try:
        two, _ = m.align_fasta_pairs([(tf, qf)], pm)
    finally:
        m.close()
    assert two == want["paf"]


def test_limits_of_the_blocked_path_are_refused_loudly(monkeypatch):
    from cactus_amd import miblast
    tf, qf = genome_like(404)
    monkeypatch.setenv("MIBLAST_BLOCK_BASES", "14000")         # every contig fits, the target needs several blocks
    m = miblast.Multi(1)
    try:
        monkeypatch.setenv("MIBLAST_BLOCK_BASES", "2000")          # smaller than a contig
        with pytest.raises(miblast.MiblastError, match="longer than"):
            m.align_fasta_pairs([(tf, qf)], miblast.params_from_args(KEG_DEFAULT))
    finally:
        m.close()


def test_blocked_equals_unblocked_at_chunk_scale(monkeypatch):
    """Blocks at a realistic scale (4 contigs of 5 Mb per file, blocks of <= 6 Mb -> 4 x 4 block pairs on two logical devices,
    batched per device): the assembled job equals the single unblocked job byte for byte (which the suite pins to the oracle at
    small sizes), for a --step=2 option set whose block origins are odd."""
    from cactus_amd import gen, miblast
    rng = np.random.default_rng(77)
    trecs, qrecs = [], []
    for k in range(4):
        t = gen.random_sequence(5_000_001 + 2 * k, rng)
        q = gen.mutate(t, rng, 0.02, 0.002)
        if k == 2:
            q = gen.revcomp(q)
        trecs.append(("id=T|chr%d" % k, gen.soft_mask(t, rng, 0.4)))
        qrecs.append(("id=Q|chr%d" % k, gen.soft_mask(q, rng, 0.4)))
    tf, qf = gen.fasta_bytes(trecs), gen.fasta_bytes([qrecs[i] for i in (2, 0, 3, 1)])
    pm = miblast.params_from_args("--step=2 --ambiguous=iupac,100,100 --ydrop=3000 --notransition".split())
    m = miblast.Multi(1)
    try:
        whole, st0 = m.align_fasta_pairs([(tf, qf)], pm)
    finally:
        m.close()
    assert whole.count(b"\n") >= 4
    monkeypatch.setenv("MIBLAST_BLOCK_BASES", "6000000")
    monkeypatch.setenv("MIBLAST_DEVICE_MAP", "0,0")
    m = miblast.Multi(2)
    try:
        blocked, st1 = m.align_fasta_pairs([(tf, qf)], pm)
    finally:
        m.close()
    assert blocked == whole
    for k in ("seed_hits", "hsps", "dp_cells", "alignments"):
        assert st0[k] == st1[k], k


def test_blocked_path_at_megabase_blocks_equals_one_oracle_run(olz, monkeypatch):
    """The blocked / multi-device path pinned to the ORACLE at a realistic block size (not only to the unblocked product): a
    multi-contig pair of 3.2 Mb cut into blocks of at most 1 Mb, dealt to two logical devices, against ONE oracle run over the
    whole files (seconds of CPU): bytes and counters."""
    import numpy as np
    from cactus_amd import gen, miblast
    rng = np.random.default_rng(77)
    trecs, qrecs = [], []
    for k in range(5):
        n = 400_000 + 90_000 * k
        t = gen.random_sequence(n, rng)
        q = gen.mutate(t, rng, 0.05, 0.004)
        if k == 1:
            q = gen.revcomp(q)
        trecs.append(("id=T|chr%d" % k, gen.soft_mask(t, rng, 0.2)))
        qrecs.append(("id=Q|chr%d" % k, gen.soft_mask(q, rng, 0.2)))
    tf, qf = gen.fasta_bytes(trecs), gen.fasta_bytes([qrecs[i] for i in (3, 1, 4, 0, 2)])
    args = "--step=2 --ambiguous=iupac,100,100 --ydrop=3500 --hspthresh=2800".split()
    pm = miblast.params_from_args(args)
    want = olz.align(tf, qf, olz.default_params(**{f: getattr(pm, f) for f, _ in pm._fields_}), details=False)
    assert want["paf"].count(b"\n") >= 5
    monkeypatch.setenv("MIBLAST_BLOCK_BASES", "1000000")
    monkeypatch.setenv("MIBLAST_DEVICE_MAP", "0,0")
    m = miblast.Multi(2)
    try:
        got, st = m.align_fasta_pairs([(tf, qf)], pm)
    finally:
        m.close()
    assert got == want["paf"]
    for k in ("seed_hits", "hits_extended", "hsps", "dp_cells", "alignments"):
        assert st[k] == want["counters"][k], k
