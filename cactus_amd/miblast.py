"""ctypes binding of libmiblast.so (include/miblast.h) -- the in-process form of the lastz /
run_kegalign replacement.  This is product code: it never imports anything from oracle/ and it
raises if the HIP library or a gfx950 device is missing (there is no CPU fallback).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmiblast.so")


class MiblastError(RuntimeError):
    pass


class Params(C.Structure):
    """miblast_params; defaults = lastz defaults."""
    _fields_ = [(n, C.c_int32) for n in ("step", "transitions", "xdrop", "ydrop", "hspthresh", "gappedthresh", "gap_open",
                                          "gap_extend", "entropy", "queryhspbest", "ambiguous_n", "gapped", "format", "markend", "queryhsplimit", "diag_hash16", "walls", "strands",
                                          "query_softmask", "step_origin", "xdrop_le", "hspbest_ties")]

    def __repr__(self):
        return "Params(" + ", ".join(f"{n}={getattr(self, n)}" for n, _ in self._fields_) + ")"


class Hsp(C.Structure):
    _fields_ = [("t_start", C.c_int32), ("q_start", C.c_int32), ("len", C.c_int32), ("score", C.c_int32),
                ("seed_t_end", C.c_int32), ("seed_q_end", C.c_int32), ("cnt", C.c_int32 * 4), ("strand", C.c_int32),
                ("q_contig", C.c_int32)]


class Aln(C.Structure):
    _fields_ = [("strand", C.c_int32), ("q_contig", C.c_int32), ("t_contig", C.c_int32), ("t_lo", C.c_int32),
                ("t_hi", C.c_int32), ("q_lo", C.c_int32), ("q_hi", C.c_int32), ("score", C.c_int32), ("dmin", C.c_int32),
                ("dmax", C.c_int32), ("anchor_t", C.c_int32), ("anchor_q", C.c_int32), ("ops_off", C.c_int64),
                ("n_ops", C.c_int64)]


class Stats(C.Structure):
    _fields_ = ([(n, C.c_int64) for n in ("seed_lookups", "seed_hits", "hits_extended", "ungapped_cols", "hsps_pre_entropy",
                                           "hsps", "anchors", "anchors_skipped", "dp_sides", "dp_cells", "dp_rows", "alignments")]
                + [(n, C.c_double) for n in ("t_index", "t_seed", "t_gapped", "t_total")]
                + [("dp_sides_run", C.c_int64), ("dp_cells_run", C.c_int64), ("gapped_rounds", C.c_int64),
                   ("seed_batches", C.c_int64), ("t_dp_kernel_ms", C.c_double), ("dp_kernel_launches", C.c_int64),
                   ("t_ungapped_kernel_ms", C.c_double), ("ungapped_kernel_launches", C.c_int64), ("t_sort_ms", C.c_double),
                   ("t_seedfill_ms", C.c_double), ("dp_rows_run", C.c_int64),
                   ("relay_accepted", C.c_int64), ("relay_rejected", C.c_int64), ("t_traceback_ms", C.c_double), ("t_merge_ms", C.c_double), ("dp_reruns", C.c_int64),
                   ("t_dp_busy_ms", C.c_double), ("relay_inline_checks", C.c_int64), ("relay_inline_continued", C.c_int64), ("seed_binned", C.c_int64)])

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class FastaPair(C.Structure):
    """miblast_fasta_pair: FASTA text of one chunk pair in memory"""
    _fields_ = [("target", C.c_char_p), ("target_len", C.c_size_t), ("query", C.c_char_p), ("query_len", C.c_size_t)]


_lib = None


def load() -> C.CDLL:
    """Loads libmiblast.so; raises MiblastError if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MiblastError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C cactus_amd/csrc`")
    lib = C.CDLL(LIB_PATH)
    vp, cp, i32, i64 = C.c_void_p, C.c_char_p, C.c_int32, C.c_int64
    P = C.POINTER
    sig = {
        "miblast_params_default": (None, [P(Params)]),
        "miblast_params_size": (C.c_size_t, []),
        "miblast_frontend_runtime_defaults": (C.c_int, [C.c_int]),
        "miblast_params_from_argv": (C.c_int, [C.c_int, P(cp), P(Params), P(cp), P(C.c_int), P(C.c_int)]),
        "miblast_device_count": (C.c_int, []),
        "miblast_set_host_threads": (C.c_int, [C.c_int]),
        "miblast_ctx_create": (C.c_int, [C.c_int, P(vp)]),
        "miblast_ctx_set_priority": (C.c_int, [vp, C.c_int]),
        "miblast_ctx_destroy": (None, [vp]),
        "miblast_seqset_from_fasta_file": (C.c_int, [vp, cp, P(vp)]),
        "miblast_seqset_from_fasta_mem": (C.c_int, [vp, cp, C.c_size_t, P(vp)]),
        "miblast_seqset_free": (None, [vp]),
        "miblast_drop_derived": (None, []),
        "miblast_seqsets_unaligned": (C.c_int, [vp, C.c_size_t, P(vp), P(cp), P(C.c_size_t), i64, i64, P(vp)]),
        "miblast_seqset_fasta": (C.c_int, [vp, P(vp), P(C.c_size_t)]),
        "miblast_seqset_n_contigs": (i32, [vp]),
        "miblast_seqset_total": (i64, [vp]),
        "miblast_seqset_name": (cp, [vp, i32]),
        "miblast_seqset_start": (i64, [vp, i32]),
        "miblast_seqset_len": (i64, [vp, i32]),
        "miblast_align": (C.c_int, [vp, vp, vp, P(Params), P(vp)]),
        "miblast_align_pairs": (C.c_int, [vp, P(vp), P(vp), C.c_size_t, P(Params), P(vp)]),
        "miblast_result_free": (None, [vp]),
        "miblast_result_paf": (vp, [vp, P(C.c_size_t)]),
        "miblast_result_stats": (P(Stats), [vp]),
        "miblast_result_hsps": (P(Hsp), [vp, P(i64)]),
        "miblast_result_alns": (P(Aln), [vp, P(i64)]),
        "miblast_result_ops": (P(C.c_uint32), [vp, P(i64)]),
        "miblast_align_files": (C.c_int, [vp, cp, cp, P(Params), C.c_int, P(Stats)]),
        "miblast_multi_create": (C.c_int, [C.c_int, P(vp)]),
        "miblast_multi_destroy": (None, [vp]),
        "miblast_multi_num_gpu": (C.c_int, [vp]),
        "miblast_multi_align_files": (C.c_int, [vp, cp, cp, P(Params), C.c_int, P(Stats)]),
        "miblast_multi_align_fasta_pairs": (C.c_int, [vp, P(FastaPair), C.c_size_t, P(Params), P(vp), P(C.c_size_t), P(Stats)]),
        "miblast_build_index": (C.c_int, [vp, vp, i32, P(P(C.c_uint32)), P(P(C.c_uint32))]),
        "miblast_free": (None, [vp]),
        "miblast_last_error": (cp, []),
        "miblast_version": (cp, []),
        "miblast_debug_device_allocs": (C.c_longlong, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)           # AttributeError here = header/library mismatch
        fn.restype, fn.argtypes = res, args
    if lib.miblast_params_size() != C.sizeof(Params):       # (the struct carries no version field: this is the check instead)
        raise MiblastError(f"miblast_params: this binding has {C.sizeof(Params)} bytes, {LIB_PATH} has {lib.miblast_params_size()}: rebuild the library or update the binding")
    _lib = lib
    return lib


EXPORTED_SYMBOLS = ("miblast_params_default", "miblast_params_size", "miblast_frontend_runtime_defaults", "miblast_params_from_argv", "miblast_device_count", "miblast_set_host_threads",
                    "miblast_ctx_create", "miblast_ctx_set_priority",
                    "miblast_ctx_destroy", "miblast_seqset_from_fasta_file", "miblast_seqset_from_fasta_mem",
                    "miblast_seqset_free", "miblast_drop_derived", "miblast_seqsets_unaligned", "miblast_seqset_fasta", "miblast_seqset_n_contigs", "miblast_seqset_total", "miblast_seqset_name",
                    "miblast_seqset_start", "miblast_seqset_len", "miblast_align", "miblast_align_pairs", "miblast_result_free",
                    "miblast_result_paf", "miblast_result_stats", "miblast_result_hsps", "miblast_result_alns",
                    "miblast_result_ops", "miblast_align_files", "miblast_multi_create", "miblast_multi_destroy", "miblast_multi_num_gpu",
                    "miblast_multi_align_files", "miblast_multi_align_fasta_pairs", "miblast_build_index", "miblast_free",
                    "miblast_last_error", "miblast_version", "miblast_debug_device_allocs")


def _check(rc: int):
    if rc != 0:
        raise MiblastError(f"miblast rc={rc}: {load().miblast_last_error().decode()}")


def default_params(**over) -> Params:
    p = Params()
    load().miblast_params_default(C.byref(p))
    for k, v in over.items():
        setattr(p, k, v)
    return p


def params_from_args(args) -> Params:
    """Parses lastz-style option strings (no file names), e.g. the <lastzArguments> sets."""
    argv = [b"lastz", b"T.fa", b"Q.fa"] + [a.encode() for a in args if a]
    arr = (C.c_char_p * len(argv))(*argv)
    p = Params()
    files = (C.c_char_p * 2)()
    ng, nt = C.c_int(), C.c_int()
    _check(load().miblast_params_from_argv(len(argv), arr, C.byref(p), files, C.byref(ng), C.byref(nt)))
    return p


def device_count() -> int:
    n = load().miblast_device_count()
    if n < 0:                                  # an invalid $MIBLAST_DEVICE_MAP is an error, not a shorter device list
        _check(n)
    return n


def device_allocs() -> int:
    """hipMalloc + hipFree calls the library has made in this process so far (miblast_debug_device_allocs)"""
    return int(load().miblast_debug_device_allocs())


def drop_derived() -> None:
    """Frees what the library keeps with resident sets (seed tables per --step, '-' strands, packed strands): made again on demand."""
    load().miblast_drop_derived()


def set_host_threads(n: int = 0) -> int:
    """Host threads beside the GPU (KegAlign's --num_threads = job.cores, local_alignment.py:58); 0 = automatic."""
    got = load().miblast_set_host_threads(int(n))
    if got < 0:
        _check(got)
    return got


@dataclass
class AlignResult:
    paf: bytes
    stats: dict
    hsps: list
    alns: list
    ops: list


class SeqSet:
    def __init__(self, ctx: "Context", handle):
        self._ctx, self._h = ctx, handle

    @property
    def total(self) -> int:
        return load().miblast_seqset_total(self._h)

    @property
    def contigs(self):
        lib = load()
        return [(lib.miblast_seqset_name(self._h, i).decode(), lib.miblast_seqset_start(self._h, i), lib.miblast_seqset_len(self._h, i))
                for i in range(lib.miblast_seqset_n_contigs(self._h))]

    def fasta_bytes(self) -> bytes:
        """FASTA text of the set (60 columns per line): what `faffy extract` would have written for a trimmed set."""
        lib = load()
        text, n = C.c_void_p(), C.c_size_t()
        _check(lib.miblast_seqset_fasta(self._h, C.byref(text), C.byref(n)))
        try:
            return C.string_at(text, n.value)
        finally:
            lib.miblast_free(text)

    def close(self):
        if self._h:
            load().miblast_seqset_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One MI355X.  Raises MiblastError when no gfx950 device is visible."""

    def __init__(self, device: int = 0):
        h = C.c_void_p()
        _check(load().miblast_ctx_create(device, C.byref(h)))
        self._h = h
        self.device = device

    def set_priority(self, level: int):
        """< 0: this context's launches yield to those of the device's other contexts (jobs nothing waits for); > 0: they go first."""
        _check(load().miblast_ctx_set_priority(self._h, int(level)))
        return self

    def seqset_from_fasta_bytes(self, data: bytes) -> SeqSet:
        h = C.c_void_p()
        _check(load().miblast_seqset_from_fasta_mem(self._h, data, len(data), C.byref(h)))
        return SeqSet(self, h)

    def seqsets_unaligned(self, queries, pafs, min_size: int, flank: int):
        """Outgroup trimming on the device (include/miblast.h miblast_seqsets_unaligned; local_alignment.py:460-499): for every
        (resident query set, PAF bytes of its alignments) the part no alignment covers, as a new resident set -- or None when
        nothing is left.  One call for all chains of a dependency level."""
        n = len(queries)
        qs = (C.c_void_p * n)(*[q._h for q in queries])
        ps = (C.c_char_p * n)(*pafs)
        ls = (C.c_size_t * n)(*[len(p) for p in pafs])
        outs = (C.c_void_p * n)()
        _check(load().miblast_seqsets_unaligned(self._h, n, qs, ps, ls, int(min_size), int(flank), outs))
        return [SeqSet(self, C.c_void_p(outs[i])) if outs[i] else None for i in range(n)]

    def seqset_from_fasta_file(self, path: str) -> SeqSet:
        h = C.c_void_p()
        _check(load().miblast_seqset_from_fasta_file(self._h, path.encode(), C.byref(h)))
        return SeqSet(self, h)

    def align(self, target: SeqSet, query: SeqSet, params: Params, details: bool = True) -> AlignResult:
        lib = load()
        r = C.c_void_p()
        _check(lib.miblast_align(self._h, target._h, query._h, C.byref(params), C.byref(r)))
        try:
            n = C.c_size_t()
            ptr = lib.miblast_result_paf(r, C.byref(n))
            paf = C.string_at(ptr, n.value) if n.value else b""
            stats = lib.miblast_result_stats(r).contents.as_dict()
            hsps = alns = ops = []
            if details:
                k = C.c_int64()
                hp = lib.miblast_result_hsps(r, C.byref(k))
                hsps = [(h.strand, h.q_contig, h.t_start, h.q_start, h.len, h.score, h.seed_t_end, h.seed_q_end, tuple(h.cnt))
                        for h in (hp[i] for i in range(k.value))]
                ap = lib.miblast_result_alns(r, C.byref(k))
                raw = [ap[i] for i in range(k.value)]
                alns = [(a.strand, a.q_contig, a.t_contig, a.t_lo, a.t_hi, a.q_lo, a.q_hi, a.score, a.dmin, a.dmax,
                         a.anchor_t, a.anchor_q, a.n_ops) for a in raw]
                op = lib.miblast_result_ops(r, C.byref(k))
                # run-length ops per alignment ((len<<2)|op), in output order
                ops = [[op[a.ops_off + j] for j in range(a.n_ops)] for a in raw] if k.value < 5_000_000 else []
            return AlignResult(paf, stats, hsps, alns, ops)
        finally:
            lib.miblast_result_free(r)

    def _unpack(self, r, details: bool) -> AlignResult:
        lib = load()
        n = C.c_size_t()
        ptr = lib.miblast_result_paf(r, C.byref(n))
        paf = C.string_at(ptr, n.value) if n.value else b""
        stats = lib.miblast_result_stats(r).contents.as_dict()
        hsps = alns = ops = []
        if details:
            k = C.c_int64()
            hp = lib.miblast_result_hsps(r, C.byref(k))
            hsps = [(h.strand, h.q_contig, h.t_start, h.q_start, h.len, h.score, h.seed_t_end, h.seed_q_end, tuple(h.cnt))
                    for h in (hp[i] for i in range(k.value))]
            ap = lib.miblast_result_alns(r, C.byref(k))
            raw = [ap[i] for i in range(k.value)]
            alns = [(a.strand, a.q_contig, a.t_contig, a.t_lo, a.t_hi, a.q_lo, a.q_hi, a.score, a.dmin, a.dmax,
                     a.anchor_t, a.anchor_q, a.n_ops) for a in raw]
            op = lib.miblast_result_ops(r, C.byref(k))
            ops = [[op[a.ops_off + j] for j in range(a.n_ops)] for a in raw] if k.value < 5_000_000 else []
        return AlignResult(paf, stats, hsps, alns, ops)

    def align_pairs(self, pairs, params: Params, details: bool = False):
        """pairs: list of (target SeqSet, query SeqSet); one batched call, gapped stages merged on the GPU."""
        lib = load()
        n = len(pairs)
        ts = (C.c_void_p * n)(*[t._h for t, _ in pairs])
        qs = (C.c_void_p * n)(*[q._h for _, q in pairs])
        rs = (C.c_void_p * n)()
        _check(lib.miblast_align_pairs(self._h, ts, qs, n, C.byref(params), rs))
        try:
            return [self._unpack(C.c_void_p(rs[i]), details) for i in range(n)]
        finally:
            for i in range(n):
                lib.miblast_result_free(C.c_void_p(rs[i]))

    def build_index(self, target: SeqSet, step: int):
        """Returns (offsets, positions) as numpy arrays (seed position table, CSR over 2^24 words)."""
        import numpy as np
        lib = load()
        off, pos = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)()
        _check(lib.miblast_build_index(self._h, target._h, step, C.byref(off), C.byref(pos)))
        try:
            o = np.ctypeslib.as_array(off, shape=((1 << 24) + 1,)).copy()
            p = np.ctypeslib.as_array(pos, shape=(max(1, int(o[-1])),)).copy()[: int(o[-1])]
        finally:
            lib.miblast_free(off)
            lib.miblast_free(pos)
        return o, p

    def close(self):
        if self._h:
            load().miblast_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Multi:
    """Several MI355X driven from this process (miblast_multi): what `run_kegalign ... --num_gpu N` is inside.  Block pairs of
    the inputs are dealt to the devices; the PAF bytes do not depend on num_gpu (include/miblast.h)."""

    def __init__(self, num_gpu: int = 1):
        h = C.c_void_p()
        _check(load().miblast_multi_create(int(num_gpu), C.byref(h)))
        self._h = h
        self.num_gpu = int(num_gpu)

    def align_files(self, target_fa: str, query_fa: str, params: Params, out_fd: int):
        st = Stats()
        _check(load().miblast_multi_align_files(self._h, target_fa.encode(), query_fa.encode(), C.byref(params), int(out_fd), C.byref(st)))
        return st.as_dict()

    def align_fasta_pairs(self, pairs, params: Params):
        """pairs: list of (target FASTA bytes, query FASTA bytes) -> (PAF bytes of all pairs in pair order, stats totals)"""
        lib = load()
        n = len(pairs)
        arr = (FastaPair * n)(*[FastaPair(t, len(t), q, len(q)) for t, q in pairs])
        out, ln, st = C.c_void_p(), C.c_size_t(), Stats()
        _check(lib.miblast_multi_align_fasta_pairs(self._h, arr, n, C.byref(params), C.byref(out), C.byref(ln), C.byref(st)))
        try:
            return (C.string_at(out, ln.value) if ln.value else b""), st.as_dict()
        finally:
            lib.miblast_free(out)

    def close(self):
        if self._h:
            load().miblast_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
