"""ctypes binding of the CPU ORACLE (oracle/liblastz_oracle.so).  TEST INFRASTRUCTURE ONLY:
importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from
cactus_amd/.  See oracle/lastz_oracle.h for the "PARITY UNPINNED" statement."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblastz_oracle.so")
CLI_PATH = os.path.join(_HERE, "oracle_lastz")


class Params(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("step", "transitions", "xdrop", "ydrop", "hspthresh", "gappedthresh", "gap_open",
                                          "gap_extend", "entropy", "queryhspbest", "ambiguous_n", "gapped", "format", "markend", "queryhsplimit", "diag_hash16", "walls", "strands",
                                          "query_softmask", "step_origin", "xdrop_le", "hspbest_ties")] + [("traceback_cells", C.c_int64)]


class SeqSetS(C.Structure):
    _fields_ = [("n_contigs", C.c_int32), ("names", C.POINTER(C.c_char_p)), ("starts", C.POINTER(C.c_int64)),
                ("lens", C.POINTER(C.c_int64)), ("total", C.c_int64), ("codes", C.POINTER(C.c_uint8))]


class Hsp(C.Structure):
    _fields_ = [("t_start", C.c_int32), ("q_start", C.c_int32), ("len", C.c_int32), ("score", C.c_int32),
                ("seed_t_end", C.c_int32), ("seed_q_end", C.c_int32), ("cnt", C.c_int32 * 4), ("strand", C.c_int32),
                ("q_contig", C.c_int32)]


class Aln(C.Structure):
    _fields_ = [("strand", C.c_int32), ("q_contig", C.c_int32), ("t_contig", C.c_int32), ("t_lo", C.c_int32),
                ("t_hi", C.c_int32), ("q_lo", C.c_int32), ("q_hi", C.c_int32), ("score", C.c_int32), ("dmin", C.c_int32),
                ("dmax", C.c_int32), ("anchor_t", C.c_int32), ("anchor_q", C.c_int32), ("ops_off", C.c_int64),
                ("n_ops", C.c_int64)]


class Counters(C.Structure):
    _fields_ = ([(n, C.c_int64) for n in ("seed_lookups", "seed_hits", "hits_extended", "ungapped_cols", "hsps_pre_entropy",
                                           "hsps", "anchors", "anchors_skipped", "dp_sides", "dp_cells", "dp_rows", "alignments")]
                + [(n, C.c_double) for n in ("t_index", "t_seed", "t_gapped", "t_total")])


class ResultS(C.Structure):
    _fields_ = [("paf", C.c_void_p), ("paf_len", C.c_size_t), ("hsps", C.POINTER(Hsp)), ("n_hsps", C.c_int64),
                ("alns", C.POINTER(Aln)), ("n_alns", C.c_int64), ("ops", C.POINTER(C.c_uint32)), ("n_ops", C.c_int64),
                ("c", Counters)]


_lib = None


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        lib = C.CDLL(LIB_PATH)
        lib.olz_params_default.argtypes = [C.POINTER(Params)]
        lib.olz_seqset_from_fasta_mem.restype = C.POINTER(SeqSetS)
        lib.olz_seqset_from_fasta_mem.argtypes = [C.c_char_p, C.c_size_t]
        lib.olz_seqset_free.argtypes = [C.POINTER(SeqSetS)]
        lib.olz_align.argtypes = [C.POINTER(SeqSetS), C.POINTER(SeqSetS), C.POINTER(Params), C.POINTER(C.POINTER(ResultS))]
        lib.olz_result_free.argtypes = [C.POINTER(ResultS)]
        lib.olz_build_index.argtypes = [C.POINTER(SeqSetS), C.c_int32, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.POINTER(C.c_uint32))]
        lib.olz_free.argtypes = [C.c_void_p]
        lib.olz_score.restype = C.c_int32
        lib.olz_score.argtypes = [C.c_uint8, C.c_uint8, C.c_int]
        _lib = lib
    return _lib


def default_params(**over) -> Params:
    p = Params()
    load().olz_params_default(C.byref(p))
    for k, v in over.items():
        setattr(p, k, v)
    return p


def align(target_fasta: bytes, query_fasta: bytes, params: Params, details: bool = True):
    """Returns dict(paf=bytes, counters=dict, hsps=[tuple], alns=[tuple], ops=[int])."""
    lib = load()
    T = lib.olz_seqset_from_fasta_mem(target_fasta, len(target_fasta))
    Q = lib.olz_seqset_from_fasta_mem(query_fasta, len(query_fasta))
    r = C.POINTER(ResultS)()
    lib.olz_align(T, Q, C.byref(params), C.byref(r))
    R = r.contents
    out = dict(paf=C.string_at(R.paf, R.paf_len) if R.paf_len else b"",
               counters={n: getattr(R.c, n) for n, _ in Counters._fields_})
    if details:
        out["hsps"] = [(h.strand, h.q_contig, h.t_start, h.q_start, h.len, h.score, h.seed_t_end, h.seed_q_end, tuple(h.cnt))
                       for h in (R.hsps[i] for i in range(R.n_hsps))]
        raw = [R.alns[i] for i in range(R.n_alns)]
        out["alns"] = [(a.strand, a.q_contig, a.t_contig, a.t_lo, a.t_hi, a.q_lo, a.q_hi, a.score, a.dmin, a.dmax,
                        a.anchor_t, a.anchor_q, a.n_ops) for a in raw]
        out["ops"] = [[R.ops[a.ops_off + j] for j in range(a.n_ops)] for a in raw] if R.n_ops < 5_000_000 else []
    lib.olz_result_free(r)
    lib.olz_seqset_free(T)
    lib.olz_seqset_free(Q)
    return out


def build_index(target_fasta: bytes, step: int):
    import numpy as np
    lib = load()
    T = lib.olz_seqset_from_fasta_mem(target_fasta, len(target_fasta))
    off, pos = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)()
    lib.olz_build_index(T, step, C.byref(off), C.byref(pos))
    o = np.ctypeslib.as_array(off, shape=((1 << 24) + 1,)).copy()
    p = np.ctypeslib.as_array(pos, shape=(max(1, int(o[-1])),)).copy()[: int(o[-1])]
    lib.olz_free(off)
    lib.olz_free(pos)
    lib.olz_seqset_free(T)
    return o, p
