"""ctypes binding of the chaining stage of libmiblast.so (include/mipaf.h): the in-process form of the `paffy`
sub-commands chain_alignments runs after the blast phase (/root/reference/src/cactus/paf/local_alignment.py:607-727).
Product code: never imports oracle/; chain, tile and trim need a gfx950 device (Context) and raise without one."""
from __future__ import annotations

import ctypes as C
from typing import Optional

from cactus_amd import miblast
from cactus_amd.miblast import MiblastError


class ChainParams(C.Structure):
    """mipaf_chain_params; defaults = the <blast> attributes of cactus_progressive_config.xml:108-111."""
    _fields_ = [("max_gap_length", C.c_int64), ("gap_open", C.c_int64), ("gap_extend", C.c_int64), ("trim_fraction", C.c_double)]


class Stats(C.Structure):
    _fields_ = ([(n, C.c_int64) for n in ("records", "groups", "query_sequences", "ops", "chain_pairs")]
                + [(n, C.c_double) for n in ("t_sort_ms", "t_chain_dp_ms", "t_tile_ms", "t_trim_ms", "t_total_s")])

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


EXPORTED_SYMBOLS = ("mipaf_set_from_mem", "mipaf_set_from_file", "mipaf_set_free", "mipaf_set_size", "mipaf_set_text", "mipaf_set_write",
                    "mipaf_invert", "mipaf_dechunk_text", "mipaf_unaligned_fasta", "mipaf_to_bed_text", "mipaf_fasta_extract_text", "mipaf_upconvert_text", "mipaf_fasta_chunk_files", "mipaf_chain_params_default", "mipaf_chain", "mipaf_tile", "mipaf_trim", "mipaf_filter",
                    "mipaf_split_by_query", "mipaf_chain_tile_trim_filter")

_bound = False


def _lib() -> C.CDLL:
    global _bound
    lib = miblast.load()
    if not _bound:
        vp, cp, P = C.c_void_p, C.c_char_p, C.POINTER
        sig = {
            "mipaf_set_from_mem": (C.c_int, [cp, C.c_size_t, P(vp)]),
            "mipaf_set_from_file": (C.c_int, [cp, P(vp)]),
            "mipaf_set_free": (None, [vp]),
            "mipaf_set_size": (C.c_int64, [vp]),
            "mipaf_set_text": (C.c_int, [vp, P(vp), P(C.c_size_t)]),
            "mipaf_set_write": (C.c_int, [vp, C.c_int]),
            "mipaf_invert": (C.c_int, [vp]),
            "mipaf_dechunk_text": (C.c_int, [cp, C.c_size_t, C.c_int32, P(vp), P(C.c_size_t)]),
            "mipaf_unaligned_fasta": (C.c_int, [cp, C.c_size_t, cp, C.c_size_t, C.c_int64, C.c_int64, P(vp), P(C.c_size_t)]),
            "mipaf_to_bed_text": (C.c_int, [cp, C.c_size_t, cp, C.c_size_t, C.c_int32, C.c_int32, C.c_int32, C.c_int64, P(vp), P(C.c_size_t)]),
            "mipaf_fasta_extract_text": (C.c_int, [cp, C.c_size_t, cp, C.c_size_t, C.c_int64, C.c_int64, C.c_int32, P(vp), P(C.c_size_t)]),
            "mipaf_upconvert_text": (C.c_int, [cp, C.c_size_t, P(cp), P(C.c_size_t), C.c_size_t, P(vp), P(C.c_size_t)]),
            "mipaf_fasta_chunk_files": (C.c_int, [cp, C.c_size_t, cp, C.c_int64, C.c_int64, P(C.c_int32)]),
            "mipaf_chain_params_default": (None, [P(ChainParams)]),
            "mipaf_chain": (C.c_int, [vp, vp, P(ChainParams), P(Stats)]),
            "mipaf_tile": (C.c_int, [vp, vp, C.c_int32, P(Stats)]),
            "mipaf_trim": (C.c_int, [vp, vp, cp, P(Stats)]),
            "mipaf_filter": (C.c_int, [vp, C.c_int64, C.c_int64, C.c_int32]),
            "mipaf_split_by_query": (C.c_int, [vp, cp, C.c_int64, P(C.c_int32)]),
            "mipaf_chain_tile_trim_filter": (C.c_int, [vp, vp, P(ChainParams), cp, C.c_int64, C.c_int32, P(Stats)]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _bound = True
    return lib


def _check(rc: int):
    if rc != 0:
        raise MiblastError(f"mipaf rc={rc}: {_lib().miblast_last_error().decode()}")


def default_chain_params(**over) -> ChainParams:
    p = ChainParams()
    _lib().mipaf_chain_params_default(C.byref(p))
    for k, v in over.items():
        setattr(p, k, v)
    return p


def dechunk_text(paf: bytes, query_only: bool = False) -> bytes:
    """`paffy dechunk [--query]` on PAF text (include/mipaf.h mipaf_dechunk_text)"""
    lib = _lib()
    out, n = C.c_void_p(), C.c_size_t()
    _check(lib.mipaf_dechunk_text(paf, len(paf), 1 if query_only else 0, C.byref(out), C.byref(n)))
    try:
        return C.string_at(out, n.value) if n.value else b""
    finally:
        lib.miblast_free(out)


def _text_out(call) -> bytes:
    lib = _lib()
    out, n = C.c_void_p(), C.c_size_t()
    _check(call(lib, C.byref(out), C.byref(n)))
    try:
        return C.string_at(out, n.value) if n.value else b""
    finally:
        lib.miblast_free(out)


def to_bed_text(paf: bytes, fasta: bytes = None, exclude_aligned: bool = False, exclude_unaligned: bool = False, include_inverted: bool = False,
                min_size: int = 0) -> bytes:
    """`paffy to_bed --binary ...` on text (include/mipaf.h mipaf_to_bed_text)"""
    return _text_out(lambda lib, o, n: lib.mipaf_to_bed_text(paf, len(paf), fasta, len(fasta) if fasta else 0, int(exclude_aligned), int(exclude_unaligned),
                                                             int(include_inverted), min_size, o, n))


def fasta_extract_text(bed: bytes, fasta: bytes, flank: int = 0, min_size: int = 1, skip_missing: bool = False) -> bytes:
    """`faffy extract -i bed fa [--flank F] [--minSize N] [--skipMissing]` on text (mipaf_fasta_extract_text)"""
    return _text_out(lambda lib, o, n: lib.mipaf_fasta_extract_text(bed, len(bed), fasta, len(fasta), flank, min_size, int(skip_missing), o, n))


def upconvert_text(paf: bytes, fastas) -> bytes:
    """`paffy upconvert -i paf trimmed_1.fa ...` on text (mipaf_upconvert_text)"""
    arr = (C.c_char_p * len(fastas))(*fastas)
    lens = (C.c_size_t * len(fastas))(*[len(f) for f in fastas])
    return _text_out(lambda lib, o, n: lib.mipaf_upconvert_text(paf, len(paf), arr, lens, len(fastas), o, n))


def fasta_chunk_files(fasta: bytes, out_dir: str, chunk_size: int, overlap: int) -> int:
    """`faffy chunk -c C -o O --dir D` (mipaf_fasta_chunk_files): writes D/chunk_<k>.fa, returns the number of files"""
    n = C.c_int32()
    _check(_lib().mipaf_fasta_chunk_files(fasta, len(fasta), out_dir.encode(), chunk_size, overlap, C.byref(n)))
    return n.value


def unaligned_fasta(paf: bytes, fasta: bytes, min_size: int, flank: int) -> bytes:
    """`paffy to_bed --excludeAligned --binary --minSize N` + `faffy extract --flank F` on text (include/mipaf.h
    mipaf_unaligned_fasta): what is left of the query file for the next outgroup of a chain"""
    lib = _lib()
    out, n = C.c_void_p(), C.c_size_t()
    _check(lib.mipaf_unaligned_fasta(paf, len(paf), fasta, len(fasta), min_size, flank, C.byref(out), C.byref(n)))
    try:
        return C.string_at(out, n.value) if n.value else b""
    finally:
        lib.miblast_free(out)


class PafSet:
    """A list of PAF records (mipaf_set).  The sub-commands change it in place and return self, so a pipeline reads like the
    reference's piped call: PafSet.from_text(t).chain(ctx).tile(ctx).trim(ctx, "0.2").filter(max_tile_level=1)."""

    def __init__(self, handle):
        self._h = handle
        self.stats: dict = {}

    @classmethod
    def from_text(cls, text) -> "PafSet":
        data = text.encode() if isinstance(text, str) else bytes(text)
        h = C.c_void_p()
        _check(_lib().mipaf_set_from_mem(data, len(data), C.byref(h)))
        return cls(h)

    @classmethod
    def from_file(cls, path: str) -> "PafSet":
        h = C.c_void_p()
        _check(_lib().mipaf_set_from_file(path.encode(), C.byref(h)))
        return cls(h)

    def __len__(self) -> int:
        return int(_lib().mipaf_set_size(self._h))

    def text_bytes(self) -> bytes:
        lib = _lib()
        buf, n = C.c_void_p(), C.c_size_t()
        _check(lib.mipaf_set_text(self._h, C.byref(buf), C.byref(n)))
        try:
            return C.string_at(buf, n.value)
        finally:
            lib.miblast_free(buf)

    def text(self) -> str:
        return self.text_bytes().decode()

    def write(self, path: str, append: bool = False):
        with open(path, "ab" if append else "wb") as f:
            f.flush()
            _check(_lib().mipaf_set_write(self._h, f.fileno()))

    def invert(self) -> "PafSet":
        _check(_lib().mipaf_invert(self._h))
        return self

    def chain(self, ctx: miblast.Context, params: Optional[ChainParams] = None) -> "PafSet":
        st = Stats()
        _check(_lib().mipaf_chain(ctx._h, self._h, C.byref(params) if params is not None else None, C.byref(st)))
        self.stats = st.as_dict()
        return self

    def tile(self, ctx: miblast.Context, hist_bins: int = 0) -> "PafSet":
        st = Stats()
        _check(_lib().mipaf_tile(ctx._h, self._h, hist_bins, C.byref(st)))
        self.stats = st.as_dict()
        return self

    def trim(self, ctx: miblast.Context, trim_identity: str) -> "PafSet":
        st = Stats()
        _check(_lib().mipaf_trim(ctx._h, self._h, str(trim_identity).encode(), C.byref(st)))
        self.stats = st.as_dict()
        return self

    def filter(self, max_tile_level: int = -1, min_chain_score: int = -1, invert: bool = False) -> "PafSet":
        _check(_lib().mipaf_filter(self._h, max_tile_level, min_chain_score, 1 if invert else 0))
        return self

    def split_by_query(self, prefix: str, min_length: int) -> int:
        n = C.c_int32()
        _check(_lib().mipaf_split_by_query(self._h, prefix.encode(), min_length, C.byref(n)))
        return n.value

    def chain_tile_trim_filter(self, ctx: miblast.Context, params: Optional[ChainParams], trim_identity: str, min_primary_chain_score: int,
                               output_secondary: bool = False) -> "PafSet":
        st = Stats()
        _check(_lib().mipaf_chain_tile_trim_filter(ctx._h, self._h, C.byref(params) if params is not None else None, str(trim_identity).encode(),
                                                   min_primary_chain_score, 1 if output_secondary else 0, C.byref(st)))
        self.stats = st.as_dict()
        return self

    def close(self):
        if self._h:
            _lib().mipaf_set_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
