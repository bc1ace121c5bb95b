"""Second, deliberately naive restatement of the chaining-stage rules used to cross-check
oracle/paffy_oracle.c on small inputs: per-column loops and exact fractions instead of the oracle's closed forms, dict-based
bookkeeping instead of sorted arrays.  Test infrastructure only."""
from __future__ import annotations

import re
from dataclasses import dataclass, field
from fractions import Fraction
from typing import List, Optional


@dataclass
class Rec:
    qn: str; ql: int; qs: int; qe: int; strand: str; tn: str; tl: int; ts: int; te: int; nm: int; nb: int; mq: int
    tp: Optional[str] = None
    AS: Optional[int] = None
    tile: int = -1
    cn: int = -1
    s1: int = -1
    ops: Optional[list] = None          # [(len, op)]
    idx: int = 0
    extra: dict = field(default_factory=dict)


def parse(text: str) -> List[Rec]:
    out = []
    for line in text.splitlines():
        if not line.strip():
            continue
        c = line.split("\t")
        r = Rec(c[0], int(c[1]), int(c[2]), int(c[3]), c[4], c[5], int(c[6]), int(c[7]), int(c[8]), int(c[9]), int(c[10]), int(c[11]))
        for t in c[12:]:
            if t.startswith("tp:A:"): r.tp = t[5]
            elif t.startswith("AS:i:"): r.AS = int(t[5:])
            elif t.startswith("tl:i:"): r.tile = int(t[5:])
            elif t.startswith("cn:i:"): r.cn = int(t[5:])
            elif t.startswith("s1:i:"): r.s1 = int(t[5:])
            elif t.startswith("cg:Z:"): r.ops = [(int(a), b) for a, b in re.findall(r"(\d+)([=XMID])", t[5:])]
        r.idx = len(out)
        out.append(r)
    return out


def fmt(r: Rec) -> str:
    c = [r.qn, r.ql, r.qs, r.qe, r.strand, r.tn, r.tl, r.ts, r.te, r.nm, r.nb, r.mq]
    if r.tp or r.tile != -1:
        c.append("tp:A:" + (r.tp if r.tp else ("S" if r.tile > 1 else "P")))
    if r.AS is not None: c.append(f"AS:i:{r.AS}")
    if r.tile != -1: c.append(f"tl:i:{r.tile}")
    if r.cn != -1: c.append(f"cn:i:{r.cn}")
    if r.s1 != -1: c.append(f"s1:i:{r.s1}")
    if r.ops is not None: c.append("cg:Z:" + "".join(f"{n}{o}" for n, o in r.ops))
    return "\t".join(str(x) for x in c) + "\n"


def dump(recs) -> str:
    return "".join(fmt(r) for r in recs)


def invert(recs):
    for r in recs:
        r.qn, r.tn = r.tn, r.qn
        r.ql, r.tl = r.tl, r.ql
        r.qs, r.ts = r.ts, r.qs
        r.qe, r.te = r.te, r.qe
        if r.ops is not None:
            ops = r.ops[::-1] if r.strand == "-" else r.ops
            r.ops = [(n, {"I": "D", "D": "I"}.get(o, o)) for n, o in ops]
    return recs


def chain(recs, max_gap, gap_open, gap_extend, trim_fraction):
    order = sorted(recs, key=lambda r: (r.qn.encode(), r.tn.encode(), r.strand == "+", r.qs, r.ts, r.idx))
    pos = {id(r): k for k, r in enumerate(order)}
    box = {}
    for r in order:
        tq = int((r.qe - r.qs) * trim_fraction / 2.0)
        tt = int((r.te - r.ts) * trim_fraction / 2.0)
        box[id(r)] = (r.qs + tq, r.qe - tq, r.ts + tt, r.te - tt)
    cs, pred = {}, {}
    for i, r in enumerate(order):
        score = r.AS if r.AS is not None else r.nm
        best, bp = 0, None
        for j in range(i):                       # ascending: a later j only wins when strictly better
            q = order[j]
            if (q.qn, q.tn, q.strand) != (r.qn, r.tn, r.strand):
                continue
            a, b = box[id(q)], box[id(r)]
            gq = b[0] - a[1]
            gt = b[2] - a[3] if r.strand == "+" else a[2] - b[3]
            if not (0 <= gq <= max_gap and 0 <= gt <= max_gap):
                continue
            val = cs[id(q)] - (gap_open + gap_extend * (gq + gt))
            if val > best:
                best, bp = val, q
        cs[id(r)] = score + best
        pred[id(r)] = bp
    nxt = 0
    for r in recs:
        r.cn = -1
    for r in sorted(order, key=lambda r: (-cs[id(r)], pos[id(r)])):
        if r.cn != -1:
            continue
        k, score = nxt, cs[id(r)]
        nxt += 1
        while r is not None and r.cn == -1:
            r.cn, r.s1 = k, score
            r = pred[id(r)]
    return sorted(order, key=lambda r: (r.cn, pos[id(r)]))


def aligned_query_bases(r):
    """query positions under = X M columns, in op order"""
    out = []
    q = r.qs if r.strand == "+" else r.qe
    for n, o in r.ops or []:
        if o == "D":
            continue
        if o in "=XM":
            out.extend(range(q, q + n) if r.strand == "+" else range(q - 1, q - 1 - n, -1))
        q += n if r.strand == "+" else -n
    return out


def tile(recs):
    order = sorted(recs, key=lambda r: (-(r.s1 if r.s1 != -1 else r.AS if r.AS is not None else r.nm), r.idx))
    counts = {}
    for r in order:
        cnt = counts.setdefault(r.qn, {})
        bases = aligned_query_bases(r)
        level = 0
        if bases:
            vals = sorted(cnt.get(b, 0) for b in bases)
            need = (len(vals) + 1) // 2                      # smallest L with 2 * #(<= L) >= n  ==  the ceil(n/2)-th smallest value
            level = vals[need - 1]
        r.tile = level + 1
        r.tp = "P" if r.tile == 1 else "S"
        for b in bases:
            cnt[b] = min(32767, cnt.get(b, 0) + 1)
    return order


def _columns(ops):
    cols = []                                                # (is_match, q_bases, t_bases) per column
    for n, o in ops:
        cols.extend([(o in "=M", 0 if o == "D" else 1, 0 if o == "I" else 1)] * n)
    return cols


def _cut(cols, x: Fraction) -> int:
    m, cut = 0, 0
    for k, (is_m, _, _) in enumerate(cols, 1):
        m += is_m
        if Fraction(m, k) < x:
            cut = k
    return cut


def _rle(cols_ops):
    out = []
    for o in cols_ops:
        if out and out[-1][1] == o:
            out[-1][0] += 1
        else:
            out.append([1, o])
    return [(n, o) for n, o in out]


def trim(recs, identity: str):
    x = Fraction(identity)
    out = []
    for r in recs:
        if r.ops is None:
            out.append(r)
            continue
        cols = _columns(r.ops)
        letters = [o for n, o in r.ops for _ in range(n)]
        pre, suf = _cut(cols, x), _cut(cols[::-1], x)
        if pre + suf >= len(cols):
            continue
        qa = sum(c[1] for c in cols[:pre]); ta = sum(c[2] for c in cols[:pre])
        qb = sum(c[1] for c in cols[len(cols) - suf:]); tb = sum(c[2] for c in cols[len(cols) - suf:])
        keep = letters[pre:len(cols) - suf]
        # adjacent equal ops of the input stay separate ops in the oracle; rebuild per original op instead of re-running RLE
        ops, at = [], 0
        for n, o in r.ops:
            lo, hi = max(at, pre), min(at + n, len(cols) - suf)
            if hi > lo:
                ops.append((hi - lo, o))
            at += n
        assert sum(n for n, _ in ops) == len(keep)
        r.ops = ops
        r.ts += ta; r.te -= tb
        if r.strand == "+":
            r.qs += qa; r.qe -= qb
        else:
            r.qe -= qa; r.qs += qb
        r.nm = sum(n for n, o in ops if o in "=M")
        r.nb = sum(n for n, _ in ops)
        out.append(r)
    return out


def filt(recs, max_tile=-1, min_chain=-1, invert_=False):
    return [r for r in recs if (((max_tile < 0 or r.tile <= max_tile) and (min_chain < 0 or r.s1 >= min_chain)) != invert_)]


# ---- random, valid PAF sets: cactus_amd.gen.random_paf (shared with bench.py) ------------------------------------------------
from cactus_amd.gen import random_paf  # noqa: E402,F401
