"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol include/miblast.h
declares, parses the reference's argv forms, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from cactus_amd import miblast
from cactus_amd.shared.common import BIN_DIR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _no_gpu():
    return miblast.device_count() == 0


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "miblast.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(miblast_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 25
    lib = miblast.load()
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert declared == set(miblast.EXPORTED_SYMBOLS)
    assert b"gfx950" in lib.miblast_version()


def test_struct_layouts_match_header():
    # 15 x int32 params + the oracle's two comparison switches + strands ; stats = 12 int64 + 4 double + extras
    assert C.sizeof(miblast.Params) == 88
    assert C.sizeof(miblast.Hsp) == 48 and C.sizeof(miblast.Aln) == 64
    assert C.sizeof(miblast.Stats) == 12 * 8 + 4 * 8 + 4 * 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 6 * 8 + 2 * 8 + 8       # + relay_accepted, relay_rejected, t_traceback_ms, t_merge_ms, dp_reruns, t_dp_busy_ms + relay_inline_checks, relay_inline_continued + seed_binned


def test_params_struct_of_header_binding_library_and_integration_stub_agree():
    """Round-5 review: INTEGRATION.md's stub declared 18 fields while the header had 22 -- miblast_params_from_argv would have written 16 bytes past
    a pasted struct.  The field list of the header, of cactus_amd/miblast.py, of INTEGRATION.md's stub and sizeof(miblast_params) as the LIBRARY
    reports it (miblast_params_size) must all agree."""
    hdr = open(os.path.join(ROOT, "include", "miblast.h")).read()
    body = hdr[hdr.index("typedef struct miblast_params {"):hdr.index("} miblast_params;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    header_fields = re.findall(r"\bint32_t\s+([a-z_0-9]+)\s*;", body)
    binding_fields = [n for n, _ in miblast.Params._fields_]
    assert header_fields == binding_fields and len(header_fields) == 22
    lib = miblast.load()
    assert lib.miblast_params_size() == C.sizeof(miblast.Params) == 4 * len(header_fields)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = doc[doc.index("class MiblastParams(C.Structure):"):]
    stub = stub[:stub.index("_lib.miblast_params_from_argv.argtypes")]
    stub_fields = re.findall(r'"([a-z_0-9]+)"', stub[stub.index("_fields_"):stub.index(")]")])
    assert stub_fields == header_fields, (stub_fields, header_fields)
    assert "%d x int32" % len(header_fields) in stub and "miblast_params_size" in stub
    # the oracle's struct is copy-compatible (tests memcpy between the two)
    ohdr = open(os.path.join(ROOT, "oracle", "lastz_oracle.h")).read()
    obody = re.sub(r"/\*.*?\*/", "", ohdr[ohdr.index("typedef struct"):], flags=re.S)
    ofields = re.findall(r"\bint32_t\s+([a-z_0-9]+)\s*;", obody[:obody.index("}")])
    assert ofields[:len(header_fields)] == header_fields


def test_polling_is_the_front_ends_choice_not_the_librarys():
    """loading libmiblast.so must not switch the whole process to busy-spinning waits.  The rule lives in
    miblast_frontend_runtime_defaults (called by bin/lastz ...); the library's constructor only sets it under MIBLAST_POLL=1."""
    code = ("import os, ctypes; os.environ.pop('HSA_ENABLE_INTERRUPT', None); os.environ.pop('MIBLAST_POLL', None); "
            "lib = ctypes.CDLL(%r); a = os.environ.get('HSA_ENABLE_INTERRUPT'); "
            "libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p; "
            "before = libc.getenv(b'HSA_ENABLE_INTERRUPT'); r = lib.miblast_frontend_runtime_defaults(64); after = libc.getenv(b'HSA_ENABLE_INTERRUPT'); "
            "r2 = lib.miblast_frontend_runtime_defaults(4); core = libc.getenv(b'HSA_DISABLE_COREDUMP_ON_EXCEPTION'); print(before, r, after, r2, core)" % miblast.LIB_PATH)
    env = {k: v for k, v in os.environ.items() if k not in ("HSA_ENABLE_INTERRUPT", "MIBLAST_POLL", "HSA_DISABLE_COREDUMP_ON_EXCEPTION")}
    out = subprocess.run([os.sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == ["None", "1", "b'0'", "0", "b'1'"], out.stdout      # (r2: a small job asks for nothing)
    out = subprocess.run([os.sys.executable, "-c", code.replace("r = lib.miblast_frontend_runtime_defaults(64)", "r = lib.miblast_frontend_runtime_defaults(8)")],
                         capture_output=True, text=True, env=env)
    assert out.stdout.split() == ["None", "0", "None", "0", "b'1'"], out.stdout
    out = subprocess.run([os.sys.executable, "-c", code.replace("os.environ.pop('MIBLAST_POLL', None); ", "")], capture_output=True, text=True, env=dict(env, MIBLAST_POLL="0"))
    assert out.stdout.split()[:3] == ["None", "0", "None"], out.stdout


def test_default_params_are_lastz_defaults():
    p = miblast.default_params()
    assert (p.step, p.transitions, p.xdrop, p.ydrop, p.hspthresh, p.gappedthresh, p.gap_open, p.gap_extend, p.entropy,
            p.queryhspbest, p.ambiguous_n, p.gapped, p.format, p.markend, p.queryhsplimit) == (1, 1, 910, 9400, 3000, -1, 400, 30, 1, 0, 1, 1, 0, 0, 0)
    assert (p.diag_hash16, p.walls, p.strands) == (0, 0, 0)
    assert [miblast.params_from_args([a]).strands for a in ("--strand=both", "--strand=plus", "--strand=minus")] == [0, 1, 2]      # lastz's --strand
    both = miblast.params_from_args(["--miblast-diag=hash16", "--miblast-walls"])       # parsed (one argv serves oracle and product) ...
    assert (both.diag_hash16, both.walls) == (1, 1)                                      # ... and refused at run time (GPU test)


def _argv(args):
    lib = miblast.load()
    argv = [a.encode() for a in args]
    arr = (C.c_char_p * len(argv))(*argv)
    p = miblast.Params()
    files = (C.c_char_p * 2)()
    ng, nt = C.c_int(-1), C.c_int(-1)
    rc = lib.miblast_params_from_argv(len(argv), arr, C.byref(p), files, C.byref(ng), C.byref(nt))
    return rc, p, [f.decode() if f else None for f in files], ng.value, nt.value, lib.miblast_last_error().decode()


def test_argv_of_run_lastz_cpu_form():
    # local_alignment.py:60-68
    rc, p, files, ng, nt, _ = _argv(["lastz", "A_0.fa[multiple][nameparse=darkspace]", "B_1.fa[nameparse=darkspace]", "--format=paf:wfmash",
                                     "--step=3", "--ambiguous=iupac,100,100", "--ydrop=3500", "--hspthresh=2600", "--gappedthresh=2800",
                                     "--queryhspbest=100000"])
    assert rc == 0 and files == ["A_0.fa[multiple][nameparse=darkspace]", "B_1.fa[nameparse=darkspace]"]
    assert (p.step, p.ydrop, p.hspthresh, p.gappedthresh, p.queryhspbest, ng, nt) == (3, 3500, 2600, 2800, 100000, 1, 1)


def test_argv_of_run_lastz_gpu_form():
    # local_alignment.py:54-58: "--num_gpu N --num_threads C" appended as separate tokens
    rc, p, files, ng, nt, _ = _argv(["run_kegalign", "A.fa", "B.fa", "--format=paf:wfmash", "--step=2", "--ambiguous=iupac,100,100",
                                     "--ydrop=3000", "--notransition", "--num_gpu", "8", "--num_threads", "32"])
    assert rc == 0 and files == ["A.fa", "B.fa"] and (p.step, p.transitions, ng, nt) == (2, 0, 8, 32)


@pytest.mark.parametrize("bad", [["--bogus"], ["--step=0"], ["--step=x"], ["--format=maf"], ["--ambiguous=weird"], ["extra.fa"],
                                 ["--num_gpu"], ["--ydrop"]])
def test_argv_rejects_what_it_does_not_implement(bad):
    rc, _, _, _, _, msg = _argv(["lastz", "A.fa", "B.fa"] + bad)
    assert rc == -1 and msg


def test_argv_needs_two_files():
    assert _argv(["lastz", "A.fa"])[0] == -1


def test_host_threads_follow_num_threads():
    # KegAlign's --num_threads = job.cores (local_alignment.py:58) bounds the host threads beside the GPU
    assert miblast.set_host_threads(3) == 3
    assert miblast.set_host_threads(1) == 1
    assert miblast.set_host_threads(1000) == 64
    auto = miblast.set_host_threads(0)
    assert 1 <= auto <= 16 and auto <= (os.cpu_count() or 1)
    env = dict(os.environ, MIBLAST_THREADS="5")
    out = subprocess.run(["python", "-c", "from cactus_amd import miblast; print(miblast.set_host_threads(0))"], capture_output=True, text=True,
                         env=env, cwd=ROOT)
    assert out.stdout.strip() == "5", out.stderr


def test_context_calls_refuse_a_null_handle():
    lib = miblast.load()
    assert lib.miblast_ctx_set_priority(None, -1) < 0
    lib.miblast_ctx_destroy(None)                                                # (a no-op, like free)


def test_no_device_means_loud_refusal_not_cpu_fallback():
    if not _no_gpu():
        pytest.skip("a GPU is visible here")
    with pytest.raises(miblast.MiblastError) as e:
        miblast.Context(0)
    assert "no CPU path" in str(e.value)


def test_multi_device_entry_refuses_without_gpu_too(monkeypatch):
    # miblast_multi (run_kegalign --num_gpu N): no device, no contexts; a device map cannot invent devices
    if not _no_gpu():
        pytest.skip("a GPU is visible here")
    monkeypatch.setenv("MIBLAST_DEVICE_MAP", "0,0")
    assert miblast.device_count() == 0
    with pytest.raises(miblast.MiblastError) as e:
        miblast.Multi(2)
    assert "no CPU path" in str(e.value)
    assert C.sizeof(miblast.FastaPair) == 32


def test_front_end_binaries_exist_and_refuse_without_gpu(tmp_path):
    for name in ("lastz", "run_kegalign"):
        exe = os.path.join(BIN_DIR, name)
        assert os.access(exe, os.X_OK), exe
        p = subprocess.run([exe, "A.fa", "B.fa", "--format=paf:wfmash", "--nonsense"], capture_output=True, cwd=tmp_path)
        assert p.returncode == 2 and p.stdout == b"" and b"unknown option" in p.stderr
    if _no_gpu():
        (tmp_path / "A.fa").write_text(">a\nACGT\n")
        p = subprocess.run([os.path.join(BIN_DIR, "lastz"), "A.fa", "A.fa", "--format=paf:wfmash"], capture_output=True, cwd=tmp_path)
        assert p.returncode == 3 and p.stdout == b""


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under cactus_amd/ may import, link or exec it."""
    pkg = os.path.join(ROOT, "cactus_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")) or f == "Makefile":
                text = open(os.path.join(base, f), errors="replace").read()
                assert "lastz_oracle" not in text and "olz_" not in text and "oracle_paffy" not in text and "paffy_oracle" not in text, os.path.join(base, f)
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), os.path.join(base, f)
    out = subprocess.run(["ldd", os.path.join(pkg, "libmiblast.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out
