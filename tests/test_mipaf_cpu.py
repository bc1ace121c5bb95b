"""Chaining stage, host side of the product (include/mipaf.h): the library exports what the header declares, PAF text round-trips,
the host-only sub-commands (invert, filter, split_file) equal the oracle byte for byte, and the three GPU sub-commands refuse to
run without a device (no CPU path)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from cactus_amd import miblast, mipaf
from cactus_amd.shared.common import BIN_DIR
from tests import pyref_paffy as ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle", "oracle_paffy")
PAFFY = os.path.join(BIN_DIR, "paffy")


def oracle(cmd, text, *args):
    p = subprocess.run([ORACLE, cmd, *args], input=text.encode(), capture_output=True)
    assert p.returncode == 0, p.stderr.decode()
    return p.stdout.decode()


def both_ways(seed):
    text = ref.random_paf(seed, n_series=6, noise=15)
    return text + ref.dump(ref.invert(ref.parse(text)))


def test_library_exports_every_symbol_of_mipaf_h():
    hdr = open(os.path.join(ROOT, "include", "mipaf.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mipaf_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(mipaf.EXPORTED_SYMBOLS) and len(declared) == 20
    lib = miblast.load()
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert C.sizeof(mipaf.ChainParams) == 32 and C.sizeof(mipaf.Stats) == 80
    p = mipaf.default_chain_params()
    assert (p.max_gap_length, p.gap_open, p.gap_extend, p.trim_fraction) == (1000000, 5000, 1, 1.0)       # xml:108-111


@pytest.mark.parametrize("seed", range(4))
def test_text_roundtrip_invert_and_filter_equal_the_oracle(seed):
    text = both_ways(seed)
    s = mipaf.PafSet.from_text(text)
    assert len(s) == len(text.splitlines())
    assert s.text() == oracle("filter", text)                       # parse + print: same tag selection and order
    assert s.invert().text() == oracle("invert", text)
    assert s.invert().text() == oracle("filter", text)              # inverting twice is the identity
    tagged = "".join(l + f"\ttp:A:{'PS'[k % 2]}\ttl:i:{1 + k % 3}\tcn:i:{k}\ts1:i:{(k * 7919) % 30000}\n" for k, l in enumerate(text.splitlines()))
    for args, kw in ((["--maxTileLevel", "1"], dict(max_tile_level=1)), (["--maxTileLevel", "2", "--invert"], dict(max_tile_level=2, invert=True)),
                     (["--minChainScore", "10000"], dict(min_chain_score=10000)), (["--minChainScore", "10000", "--invert"], dict(min_chain_score=10000, invert=True)),
                     (["--maxTileLevel", "1", "--minChainScore", "5000"], dict(max_tile_level=1, min_chain_score=5000))):
        assert mipaf.PafSet.from_text(tagged).filter(**kw).text() == oracle("filter", tagged, *args), args


def test_front_end_invert_filter_split_match_the_oracle(tmp_path):
    text = both_ways(11)
    src = tmp_path / "in.paf"
    src.write_text(text)
    for cmd in (["invert"], ["filter", "--maxTileLevel", "1"]):
        got = subprocess.run([PAFFY, *cmd, "--inputFile", str(src)], capture_output=True)
        assert got.returncode == 0 and got.stdout.decode() == oracle(cmd[0], text, *cmd[1:])
    piped = subprocess.run([PAFFY, "invert"], input=text.encode(), capture_output=True)            # stdin, as inside cactus_call pipes
    assert piped.stdout.decode() == oracle("invert", text)
    for who, exe in (("o", ORACLE), ("p", PAFFY)):
        subprocess.run([exe, "split_file", "--inputFile", str(src), "--query", "--prefix", str(tmp_path / f"{who}_split_"), "--minLength", "150000",
                        "--logLevel", "INFO"], check=True)
    o_parts = sorted(p.name[2:] for p in tmp_path.glob("o_split_*.paf"))
    assert o_parts == sorted(p.name[2:] for p in tmp_path.glob("p_split_*.paf")) and len(o_parts) >= 2
    for name in o_parts:
        assert (tmp_path / ("o_" + name)).read_text() == (tmp_path / ("p_" + name)).read_text()


@pytest.mark.parametrize("bad", ["q\t10\t0\t5\t+\tt\t10\t0\n", "q\t10\tx\t5\t+\tt\t10\t0\t5\t5\t5\t255\n", "q\t10\t0\t5\t*\tt\t10\t0\t5\t5\t5\t255\n",
                                 "q\t10\t0\t5\t+\tt\t10\t0\t5\t5\t5\t255\tcg:Z:5Q\n", "q\t10\t0\t5\t+\tt\t10\t0\t5\t5\t5\t255\tcg:Z:0=\n"])
def test_malformed_paf_is_an_error_not_a_guess(bad):
    with pytest.raises(miblast.MiblastError) as e:
        mipaf.PafSet.from_text(bad)
    assert "PAF line 1" in str(e.value)


def test_gpu_sub_commands_have_no_cpu_path(tmp_path):
    if miblast.device_count() > 0:
        pytest.skip("a GPU is visible here")
    lib = mipaf._lib()
    s = mipaf.PafSet.from_text(both_ways(1))
    before = s.text()
    assert lib.mipaf_chain(None, s._h, None, None) == -3                                           # MIBLAST_ENODEV
    assert lib.mipaf_tile(None, s._h, 0, None) == -3 and lib.mipaf_trim(None, s._h, b"0.2", None) == -3
    assert lib.mipaf_chain_tile_trim_filter(None, s._h, None, b"0.2", 10000, 0, None) == -3
    assert b"no CPU path" in lib.miblast_last_error() and s.text() == before
    for cmd in (["chain"], ["tile"], ["trim", "--trimIdentity", "0.2"]):
        p = subprocess.run([PAFFY, *cmd], input=before.encode(), capture_output=True)
        assert p.returncode == 3 and p.stdout == b"" and b"no CPU path" in p.stderr
    p = subprocess.run([PAFFY, "chain", "--frobnicate", "1"], input=b"", capture_output=True)
    assert p.returncode == 2 and b"unknown option" in p.stderr


def test_large_text_is_parsed_in_chunks_with_the_same_result():
    # > 512 KiB per worker chunk: the threaded parse / format paths; the oracle reads the same text in one pass
    text = ref.random_paf(99, n_series=120, per_series=(20, 60), n_q=3, n_t=3, contig_len=5_000_000, noise=1500, ragged=False)
    text += mipaf.PafSet.from_text(text).invert().text()
    assert len(text) > 2_000_000
    s = mipaf.PafSet.from_text(text)
    assert len(s) == len(text.splitlines())
    assert s.text() == oracle("filter", text)
    assert s.invert().text() == oracle("invert", text)
    lines = text.splitlines(keepends=True)
    k = len(lines) * 3 // 4
    lines[k] = lines[k].replace("\t+\t", "\t?\t").replace("\t-\t", "\t?\t")
    with pytest.raises(miblast.MiblastError) as e:
        mipaf.PafSet.from_text("".join(lines))
    assert f"PAF line {k + 1}:" in str(e.value)


def test_front_end_dechunks_and_hands_foreign_sub_commands_to_the_next_paffy(tmp_path):
    from cactus_amd.paf import chunking
    # `paffy dechunk -i X [--query]` (local_alignment.py:352, :515): same text as cactus_amd.paf.chunking, every other column kept
    lines = [f"id=Q|c{k}|{1000 + k}|{100 * k}\t50\t5\t20\t{'+-'[k % 2]}\tid=T|x|2000|{300 + k}\t80\t7\t30\t10\t15\t255\tAS:i:{k}\tzz:Z:keep\n" for k in range(5)]
    src = tmp_path / "c.paf"
    src.write_text("".join(lines))
    for extra, query_only in (([], False), (["--query", "--logLevel", "INFO"], True)):
        p = subprocess.run([PAFFY, "dechunk", "-i", str(src), *extra], capture_output=True)
        assert p.returncode == 0 and p.stdout.decode() == "".join(chunking.paf_dechunk_line(l, query_only) for l in lines)
    assert subprocess.run([PAFFY, "dechunk"], input=b"q\t1\t0\t1\t+\tt|9|0\t1\t0\t1\t1\t1\t255\n", capture_output=True).returncode == 1
    # any sub-command this front end does not provide goes to the next paffy on PATH (so <repo>/bin first on PATH hides nothing)
    other = tmp_path / "elsewhere"
    other.mkdir()
    (other / "paffy").write_text("#!/bin/sh\necho real paffy got: \"$@\"\nexit 7\n")
    (other / "paffy").chmod(0o755)
    env = dict(os.environ, PATH=BIN_DIR + os.pathsep + str(other) + os.pathsep + os.environ.get("PATH", ""))
    p = subprocess.run(["paffy", "view", "a.fa", "b.fa", "-i", "x.paf"], capture_output=True, env=env)
    assert p.returncode == 7 and p.stdout == b"real paffy got: view a.fa b.fa -i x.paf\n"
    assert b"`view` handed to " + os.path.realpath(other / "paffy").encode() in p.stderr          # which implementation ran is on record
    # ... and so do the sub-commands it does provide, unless MIPAF_NATIVE=1 asks for the MI355X implementation
    p = subprocess.run(["paffy", "chain", "-i", "x.paf"], capture_output=True, env=dict(env, MIPAF_NATIVE="0"))
    assert p.returncode == 7 and b"`chain` handed to " in p.stderr and b"MIPAF_NATIVE=1" in p.stderr
    p = subprocess.run(["paffy", "chain", "-i", "x.paf"], capture_output=True, env=dict(env, MIPAF_NATIVE="0", MIPAF_QUIET="1"))
    assert p.returncode == 7 and p.stderr == b""
    p = subprocess.run([PAFFY, "add_mismatches", "-i", "x.paf"], capture_output=True, env=dict(os.environ, PATH=BIN_DIR + os.pathsep + "/usr/bin:/bin"))
    assert p.returncode == 2 and b"no other paffy is on PATH" in p.stderr
    # to_bed and upconvert are this front end's own since round 4 (mp_text.cpp): the form Cactus does not use is refused, not handed on
    p = subprocess.run([PAFFY, "to_bed", "-i", str(src)], capture_output=True, env=dict(os.environ, PATH=BIN_DIR + os.pathsep + "/usr/bin:/bin"))
    assert p.returncode == 2 and b"--binary" in p.stderr
