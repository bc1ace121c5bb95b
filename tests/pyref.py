"""A second, independent restatement of SURVEY.md Appendix A.10 in pure Python (small inputs only).
It shares no code with oracle/lastz_oracle.c: agreement between the two pins the C oracle to the
written spec rather than to its own bugs.  Deliberately naive (dicts, lists, full recomputation)."""
from __future__ import annotations

import math

CARE = (0, 1, 2, 4, 7, 8, 11, 13, 15, 16, 17, 18)
HOX = {"AA": 91, "CC": 100, "GG": 100, "TT": 91, "AC": -114, "AG": -31, "AT": -123, "CG": -125, "CT": -31, "GT": -114}
NEG = -(10 ** 9)


def S(a, b):
    a, b = a.upper(), b.upper()
    if a not in "ACGT" or b not in "ACGT":
        return -100
    return HOX[a + b] if a + b in HOX else HOX[b + a]


def revcomp(s):
    m = {"A": "T", "C": "G", "G": "C", "T": "A", "a": "t", "c": "g", "g": "c", "t": "a"}
    return "".join(m.get(c, c) for c in reversed(s))


def window_ok(s, p):
    return p + 19 <= len(s) and all(c in "ACGT" for c in s[p:p + 19])


def word(s, p):
    w = 0
    for k in CARE:
        w = w * 4 + "ACGT".index(s[p + k])
    return w


def ungapped(T, Q, t_end, q_end, xdrop):
    run = bestL = bl = 0
    k = 1
    while t_end - k >= 0 and q_end - k >= 0:
        run += S(T[t_end - k], Q[q_end - k])
        if run > bestL:
            bestL, bl = run, k
        elif run < bestL - xdrop:
            break
        k += 1
    run = bestR = br = 0
    k = 0
    while t_end + k < len(T) and q_end + k < len(Q):
        run += S(T[t_end + k], Q[q_end + k])
        if run > bestR:
            bestR, br = run, k + 1
        elif run < bestR - xdrop:
            break
        k += 1
    return (t_end - bl, q_end - bl, bl + br, bestL + bestR)


def entropy_ok(T, Q, h, K):
    cnt = {}
    for k in range(h[2]):
        a, b = T[h[0] + k].upper(), Q[h[1] + k].upper()
        if a == b and a in "ACGT":
            cnt[a] = cnt.get(a, 0) + 1
    n = sum(cnt.values())
    if n == 0:
        return False
    H = -sum(c / n * math.log(c / n) for c in cnt.values()) / math.log(4.0)
    return h[3] * H >= K


def one_sided(a, b, O, E, Y):
    """a: target bases along columns (a[0] is column 1), b: query bases along rows.  Returns
    (best, bi, bj, cells, ops) with ops in walk-back order ('M','I','D')."""
    na, nb = len(a), len(b)
    best, bi, bj = 0, 0, 0
    C = {(0, 0): 0}
    D, I, tr = {}, {}, {(0, 0): (3, 0, 0)}
    R0 = min(na, (Y - O) // E) if Y >= O else 0
    for j in range(1, R0 + 1):
        C[(0, j)] = -(O + j * E)
        tr[(0, j)] = (2, 0, 1 if j >= 2 else 0)
    cells = R0 + 1
    LY, RY = 0, R0 + 1
    for i in range(1, nb + 1):
        Iv, Cleft = NEG, NEG
        first = last = None
        j = LY
        while j <= na:
            diag = C.get((i - 1, j - 1), NEG) + S(a[j - 1], b[i - 1]) if (LY <= j - 1 < RY) else NEG
            if j < RY:
                ext, opn = D.get((i - 1, j), NEG) - E, C.get((i - 1, j), NEG) - O - E
                Dv, Dext = (ext, 1) if ext >= opn else (opn, 0)
            else:
                Dv, Dext = NEG, 0
            ext, opn = Iv - E, Cleft - O - E
            Iv, Iext = (ext, 1) if ext >= opn else (opn, 0)
            if diag >= Dv and diag >= Iv:
                Cv, src = diag, 0
            elif Dv >= Iv:
                Cv, src = Dv, 1
            else:
                Cv, src = Iv, 2
            cells += 1
            if Cv > best:
                best, bi, bj = Cv, i, j
            alive = Cv >= best - Y
            C[(i, j)] = Cv if alive else NEG
            D[(i, j)] = Dv
            Cleft = C[(i, j)]
            tr[(i, j)] = (src, Dext, Iext)
            if alive:
                first = j if first is None else first
                last = j
            elif j >= RY:
                break
            j += 1
        if first is None:
            break
        LY, RY = first, last + 1
    ops, i, j, st = [], bi, bj, 0
    while i > 0 or j > 0:
        src, dext, iext = tr[(i, j)]
        if st == 0:
            if src == 0:
                ops.append("M"); i -= 1; j -= 1
            elif src == 1:
                st = 1
            elif src == 2:
                st = 2
            else:
                break
        elif st == 1:
            ops.append("I"); st = 1 if dext else 0; i -= 1
        else:
            ops.append("D"); st = 2 if iext else 0; j -= 1
    return best, bi, bj, cells, ops


def align(T, Q, step=1, transitions=True, xdrop=910, ydrop=9400, K=3000, L=None, O=400, E=30, entropy=True):
    """Single-contig target and query.  Returns (paf_lines, counters)."""
    L = K if L is None else L
    index = {}
    for p in range(0, len(T) - 18, step):
        if window_ok(T, p):
            index.setdefault(word(T, p), []).append(p)
    out, ctr = [], dict(seed_hits=0, hits_extended=0, hsps=0, dp_cells=0)
    for strand in (0, 1):
        Qs = Q if strand == 0 else revcomp(Q)
        extent, hsps = {}, []
        for q in range(len(Qs) - 18):
            if not window_ok(Qs, q):
                continue
            w = word(Qs, q)
            variants = [w] + ([w ^ (2 << (2 * (11 - k))) for k in range(12)] if transitions else [])
            for v in variants:
                for p in reversed(index.get(v, [])):
                    ctr["seed_hits"] += 1
                    t_end, q_end = p + 19, q + 19
                    d = t_end - q_end
                    if q_end <= extent.get(d, -1):
                        continue
                    h = ungapped(T, Qs, t_end, q_end, xdrop)
                    ctr["hits_extended"] += 1
                    extent[d] = h[1] + h[2]
                    if h[3] >= K and (not entropy or entropy_ok(T, Qs, h, K)):
                        hsps.append(h)
        ctr["hsps"] += len(hsps)
        anchors = []
        for (ts, qs, ln, sc) in hsps:
            if ln <= 31:
                off = ln // 2
            else:
                sums = [sum(S(T[ts + c + k], Qs[qs + c + k]) for k in range(31)) for c in range(ln - 30)]
                off = sums.index(max(sums)) + 15
            anchors.append((-sc, ts + off, qs + off))
        anchors.sort()
        kept = []
        for (nsc, t, q) in anchors:
            if any(a[0] <= t < a[1] and a[2] <= q < a[3] and a[4] <= t - q <= a[5] for a in kept):
                continue
            Rb, Ri, Rj, Rc, Rops = one_sided(T[t:], Qs[q:], O, E, ydrop)
            Lb, Li, Lj, Lc, Lops = one_sided(T[:t][::-1], Qs[:q][::-1], O, E, ydrop)
            ctr["dp_cells"] += Rc + Lc
            if Rb + Lb < L:
                continue
            cols = Lops + Rops[::-1]
            tt, qq = t - Lj, q - Li
            ops, ds = [], []
            for o in cols:
                if o == "M":
                    a, b = T[tt].upper(), Qs[qq].upper()
                    ops.append("=" if a == b and a in "ACGT" else "X")
                    ds.append(tt - qq); tt += 1; qq += 1
                elif o == "I":
                    ops.append("I"); qq += 1
                else:
                    ops.append("D"); tt += 1
            kept.append((t - Lj, t + Rj, q - Li, q + Ri, min(ds), max(ds)))
            cig, n = "", 0
            for k, o in enumerate(ops):
                n += 1
                if k + 1 == len(ops) or ops[k + 1] != o:
                    cig += "%d%s" % (n, o); n = 0
            qs_, qe_ = q - Li, q + Ri
            if strand:
                qs_, qe_ = len(Q) - qe_, len(Q) - qs_
            out.append((strand, t - Lj, t + Rj, qs_, qe_, Rb + Lb, ops.count("="), len(ops), cig))
    return out, ctr
