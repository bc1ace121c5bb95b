"""The blast phase of one progressive-Cactus run as a list of lastz calls: which genome pairs are aligned, in what order and
with which option set.  Host mirror of the scheduling half of the hot path:

    get_distances / get_event_pairs      /root/reference/src/cactus/paf/paf.py:29-71
    ancestor up-weighting of distances   /root/reference/src/cactus/progressive/progressive_decomposition.py:208-241
    make_paf_alignments                  /root/reference/src/cactus/paf/local_alignment.py:751-858   (ingroup pairs, outgroups nearest first)
    make_ingroup_to_outgroup_alignments_0..3  :421-526   (align to the nearest outgroup, keep what stayed unaligned, go on)

The reference runs these as Toil jobs over files; here the same data flow is a plain function over FASTA / PAF bytes with the
aligner passed in, so that the MI355X path (batched miblast calls) and the CPU oracle (tests, bench.py's cpu_baseline leg) run
the IDENTICAL list of calls and can be diffed call by call.  Which outgroups a node gets is Cactus's own policy
(progressive_decomposition.compute_outgroups -- orchestration, out of scope): the stand-in takes the <= max_num_outgroups
(cactus_progressive_config.xml:543) nearest leaves outside the node's subtree.

Data formats either side of a call come from cactus_amd.paf.chunking: NAME|SEQLEN|START sub-sequence names
(faffy extract), `paffy dechunk --query`, `paffy invert`.
"""
from __future__ import annotations

import threading
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from cactus_amd.paf import chunking

# /root/reference/examples/evolverMammals.txt:1 with the internal nodes named as progressive Cactus names them
EVOLVER_MAMMALS_TREE = ("((simHuman_chr6:0.144018,(simMouse_chr6:0.084509,simRat_chr6:0.091589)mr:0.271974)Anc1:0.020593,"
                        "(simCow_chr6:0.18908,simDog_chr6:0.16303)Anc2:0.032898)Anc0;")
# /root/reference/examples/evolverPrimates.txt:1, verbatim (the reference names cb and hcb itself; the unnamed root becomes Anc0).
# Every pair of this tree is closer than divergence "one" (0.05), also after the ancestor up-weighting: set "one" everywhere.
EVOLVER_PRIMATES_TREE = "(simOrang:0.00993,((simChimp:0.00272,simHuman:0.00269)cb:0.00415,simGorilla:0.00644)hcb:0.00046);"


@dataclass(eq=False)
class Node:
    iD: str
    distance: float = 0.0
    children: List["Node"] = field(default_factory=list)

    def subtree(self) -> List["Node"]:
        out = [self]
        for c in self.children:
            out += c.subtree()
        return out

    def leaves(self) -> List["Node"]:
        return [n for n in self.subtree() if not n.children]


def parse_newick(text: str) -> Node:
    """Binary (or any) rooted tree with named leaves and optionally named internal nodes; unnamed ones become Anc<k> in
    breadth-first order like Cactus does."""
    pos = 0
    text = text.strip()

    def node() -> Node:
        nonlocal pos
        kids = []
        if text[pos] == "(":
            pos += 1
            while True:
                kids.append(node())
                if text[pos] == ",":
                    pos += 1
                    continue
                assert text[pos] == ")", text[pos:]
                pos += 1
                break
        start = pos
        while pos < len(text) and text[pos] not in ",():;":
            pos += 1
        name = text[start:pos]
        dist = 0.0
        if pos < len(text) and text[pos] == ":":
            pos += 1
            start = pos
            while pos < len(text) and text[pos] not in ",();":
                pos += 1
            dist = float(text[start:pos])
        return Node(name, dist, kids)

    root = node()
    k, queue = 0, [root]
    while queue:
        n = queue.pop(0)
        if n.children and not n.iD:
            n.iD = "Anc%d" % k
            k += 1
        queue += n.children
    return root


def get_node(tree: Node, name: str) -> Node:
    return [n for n in tree.subtree() if n.iD == name][0]


def get_distances(root: Node) -> Dict[Tuple[str, str], float]:
    """Path length between every pair of nodes of the tree (paf.py:29-58), keyed by node names."""
    d: Dict[Tuple[str, str], float] = {}

    def walk(n: Node):
        d[(n.iD, n.iD)] = 0.0
        for c in n.children:
            walk(c)
            for x in c.subtree():
                v = d[(x.iD, c.iD)] + c.distance
                d[(x.iD, n.iD)] = d[(n.iD, x.iD)] = v
        for i, a in enumerate(n.children):
            for b in n.children[i + 1:]:
                for x in a.subtree():
                    for y in b.subtree():
                        v = d[(x.iD, a.iD)] + d[(y.iD, b.iD)] + a.distance + b.distance
                        d[(x.iD, y.iD)] = d[(y.iD, x.iD)] = v

    walk(root)
    return d


def get_event_pairs(tree: Node, events: Sequence[Node]):
    """paf.py:61-71: every pair of the given events with their distance in the tree, in list order."""
    d = get_distances(tree)
    for i in range(len(events)):
        for j in range(i + 1, len(events)):
            yield events[i], events[j], d[(events[i].iD, events[j].iD)]


def ancestor_scaled_tree(root: Node, max_div: float) -> Node:
    """progressive_decomposition.py:231-239 (upweightAncestorDistances="1", cactus_progressive_config.xml:11): the height of an
    ancestor (longest path to a leaf below it) is added to the branch above it, up to max_div (= divergence "five"); branches
    already longer than max_div are left alone.  Returns a scaled copy."""
    def height(n: Node) -> float:
        return max((c.distance + height(c) for c in n.children), default=0.0)

    def copy(n: Node, is_root: bool) -> Node:
        dist = n.distance
        if not is_root and n.children and dist < max_div:
            dist = min(max_div, dist + height(n))
        return Node(n.iD, dist, [copy(c, False) for c in n.children])

    return copy(root, True)


@dataclass
class Call:
    """One lastz invocation of the phase = one run_lastz job (local_alignment.py:29-97; single chunk per genome at evolver sizes)"""
    node: str                  # internal node whose blast job this call belongs to
    kind: str                  # "ingroup" (I1 x I2) or "outgroup" (ingroup -> k-th outgroup)
    target: str                # event_a: file A of run_lastz
    query: str                 # event_b: file B
    distance: float
    level: int                 # 0: no dependency; k: needs the level k-1 call of the same (node, ingroup) chain
    chain: Optional[Tuple[str, str]] = None        # (node, ingroup) for outgroup calls


def blast_phase_calls(tree: Node, max_div: float = 0.25, max_outgroups: int = 3) -> List[Call]:
    """Every lastz call of the blast phase of a whole progressive run over `tree`, node by node in post-order
    (cactus_progressive.py:157-177 walks the same way): per internal node one call per ingroup pair
    (make_paf_alignments :806-813) and, per ingroup, a chain of calls to the node's outgroups nearest first (:818-835,
    :421-526).  Distances (and with them the option sets) come from the ancestor-scaled tree."""
    scaled = ancestor_scaled_tree(tree, max_div)
    dist = get_distances(scaled)
    calls: List[Call] = []

    def post(n: Node):
        for c in n.children:
            post(c)
        if not n.children:
            return
        ingroups = list(n.children)
        sn = get_node(scaled, n.iD)
        for a, b, dab in get_event_pairs(sn, [get_node(scaled, c.iD) for c in ingroups]):
            calls.append(Call(n.iD, "ingroup", a.iD, b.iD, dab, 0))
        inside = {x.iD for x in n.subtree()}
        outgroups = sorted((l for l in tree.leaves() if l.iD not in inside), key=lambda l: dist[(n.iD, l.iD)])[:max_outgroups]
        for ing in ingroups:
            for k, og in enumerate(outgroups):
                # make_chunked_alignments(outgroup, ..., ingroup, ...): the outgroup is file A (target), the ingroup file B (query)
                calls.append(Call(n.iD, "outgroup", og.iD, ing.iD, dist[(ing.iD, og.iD)], k, (n.iD, ing.iD)))

    post(tree)
    return calls


# ---- sub-sequence bookkeeping of the outgroup chain, on bytes ------------------------------------------------------------------
def parse_fasta_bytes(data: bytes) -> List[Tuple[str, np.ndarray]]:
    recs: List[Tuple[str, np.ndarray]] = []
    for block in data.split(b">")[1:]:
        nl = block.find(b"\n")
        header = block[:nl if nl >= 0 else len(block)].decode()
        body = block[nl + 1:] if nl >= 0 else b""
        recs.append((header.split()[0] if header.split() else header, np.frombuffer(body.replace(b"\n", b"").replace(b"\r", b""), dtype=np.uint8)))
    return recs


_PARSED: Dict[bytes, List[Tuple[str, np.ndarray]]] = {}


def parsed_records(fa: bytes, keep: bool = False) -> List[Tuple[str, np.ndarray]]:
    """records of a FASTA text; the whole-genome files of a run are parsed once (keep=True) like they are uploaded once"""
    got = _PARSED.get(fa)
    if got is None:
        got = parse_fasta_bytes(fa)
        if keep:
            _PARSED[fa] = got
    return got


def unaligned_fasta(paf: bytes, query_fa: bytes, min_size: int, flank: int) -> bytes:
    """`paffy to_bed --excludeAligned --minSize N` + `faffy extract --flank F` (local_alignment.py:460-475): the parts of the QUERY
    file no alignment of `paf` covers, at least min_size long, widened by flank, as records NAME|SEQLEN|START."""
    from cactus_amd import mipaf
    return mipaf.unaligned_fasta(paf, query_fa, min_size, flank)


def unaligned_fasta_py(paf: bytes, query_fa: bytes, min_size: int, flank: int) -> bytes:
    """the same through the Python cores of cactus_amd.paf.chunking (what the file-based front ends use; the native text code
    is checked against this in tests/test_blast_phase_cpu.py)"""
    from cactus_amd import gen
    recs = parsed_records(query_fa)
    bed = chunking.unaligned_intervals(paf.decode().splitlines(), [(n, len(s)) for n, s in recs], min_size)
    return gen.fasta_bytes(chunking.extract_records(bed, recs, flank))


def dechunk_query(paf: bytes) -> bytes:
    """`paffy dechunk --query` (local_alignment.py:515), through the library's host-side text code (mipaf_dechunk_text)"""
    from cactus_amd import mipaf
    return mipaf.dechunk_text(paf, query_only=True)


def invert(paf: bytes) -> bytes:
    """`paffy invert` (local_alignment.py:411-418) through the library's host-side PAF code (include/mipaf.h mipaf_invert: the
    cigars of whole-chunk alignments run to megabytes, which is no work for a per-op Python loop)"""
    if not paf.strip():
        return b""
    from cactus_amd import mipaf
    s = mipaf.PafSet.from_text(paf)
    try:
        return s.invert().text_bytes()
    finally:
        s.close()


AlignBatch = Callable[[List[Tuple[bytes, bytes]], str], List[bytes]]


def as_fasta(query) -> bytes:
    """FASTA bytes of a query of the phase: the bytes themselves, or the text of a resident set of the aligner's (a handle made by
    align_batch.trim_resident; anything with fasta_bytes())"""
    return query if isinstance(query, (bytes, bytearray)) else query.fasta_bytes()


_POOL = None


def _phase_pool():
    """Worker threads shared by every run of the phase in this process (starting eight threads per run cost a few ms each time)."""
    global _POOL
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=8, thread_name_prefix="blast-phase")
    return _POOL


def run_blast_phase(genomes: Dict[str, bytes], calls: Sequence[Call], option_string: Callable[[float], str], align_batch: AlignBatch,
                    trim_min_size: int = 100, trim_flanking: int = 100, on_call=None) -> Dict[str, Dict[str, bytes]]:
    """Runs every call of the phase.  genomes: event name -> FASTA bytes.  align_batch(pairs, lastz_option_string) aligns a list
    of (target FASTA, query FASTA) with ONE option set and returns their PAF bytes in order -- calls of one dependency level
    that share an option set are handed over together (they are independent Toil jobs in the reference; a GPU runs them as one
    batched call).  Returns per node {"ingroup": PAF of the ingroup pairs, "outgroup": the inverted ingroup->outgroup PAF
    (make_ingroup_to_outgroup_alignments_0)}.  on_call(call, target_fa, query_fa, paf) sees every single call (parity checks).
    An aligner that keeps its sequences resident may offer align_batch.trim_resident(items, min_size, flank) -- items = [(query
    of the chain's previous call, its PAF)] -- and return, per item, a handle of its own for what is left of the query (falsy:
    nothing); such a handle comes back as the query of the chain's next pair and must offer fasta_bytes() for on_call."""
    raw: Dict[int, bytes] = {}
    query_fa: Dict[int, bytes] = {}
    current: Dict[Tuple[str, str], bytes] = {}                # (node, ingroup) -> what is left of the ingroup for the next outgroup
    last_paf: Dict[Tuple[str, str], Tuple[bytes, bytes]] = {}       # (node, ingroup) -> (query FASTA, raw PAF) of the previous level
    pool = _phase_pool()                              # the chains of a level are independent jobs (numpy / the library release the GIL)
    finished: Dict[int, object] = {}

    def chain_share(paf: bytes, depth: int) -> bytes:
        for _ in range(depth):
            paf = dechunk_query(paf) if paf else paf
        return invert(paf)
    last_level = max(c.level for c in calls)
    gate = threading.BoundedSemaphore(max(1, int(getattr(align_batch, "concurrent", 1))))
    free_jobs = []                                    # (option set, calls, future): jobs nothing waits for, collected at the end

    def gated(pairs, opts, free=False):
        with gate:
            # (an aligner may serve the jobs nothing waits for differently -- align_batch.background: on contexts whose launches
            #  yield to those of the chains)
            return (getattr(align_batch, "background", None) or align_batch)(pairs, opts) if free else align_batch(pairs, opts)

    def take(opts, idx, outs):
        for i, paf in zip(idx, outs):
            raw[i] = paf
            if calls[i].chain is not None:
                last_paf[calls[i].chain] = (query_fa[i], paf)
                # its share of the chain's final file is a job of its own (dechunk and invert work record by record), started
                # now: it runs beside the next level's calls instead of after the last one
                finished[i] = pool.submit(chain_share, paf, calls[i].level)
            if on_call is not None:
                on_call(calls[i], genomes[calls[i].target], as_fasta(query_fa[i]), paf)

    def run_level(level):
        todo: List[int] = []
        trimmed = {}
        if level > 0:
            idx = [i for i, c in enumerate(calls) if c.level == level and c.kind != "ingroup"]
            trim_resident = getattr(align_batch, "trim_resident", None)
            if trim_resident is not None:
                # the aligner keeps the sequences on its device: what is left of every chain's ingroup is cut out there, all chains
                # of the level in one call (miblast_seqsets_unaligned); a query is then a handle of the aligner's, not FASTA bytes
                live = [i for i in idx if last_paf[calls[i].chain][0]]                        # (a chain with nothing left stays empty)
                got = {}
                if live:
                    with gate:                        # (a trimming call occupies a context of the aligner like any other call)
                        got = dict(zip(live, trim_resident([last_paf[calls[i].chain] for i in live], trim_min_size, trim_flanking)))
                outs = [got.get(i) for i in idx]
            else:
                outs = pool.map(lambda i: unaligned_fasta(last_paf[calls[i].chain][1], last_paf[calls[i].chain][0], trim_min_size, trim_flanking), idx)
            trimmed = dict(zip(idx, outs))                                                      # make_ingroup_to_outgroup_alignments_2
        for i, c in enumerate(calls):
            if c.level != level:
                continue
            if c.kind == "ingroup" or level == 0:
                query_fa[i] = genomes[c.query]
            else:
                left = trimmed[i]
                if not left:
                    raw[i] = b""
                    query_fa[i] = b""
                    last_paf[c.chain] = (b"", b"")
                    continue
                query_fa[i] = left
            todo.append(i)
        by_opts: Dict[str, List[int]] = {}
        for i in todo:
            by_opts.setdefault(option_string(calls[i].distance), []).append(i)
        # the option sets of a level are independent jobs as well: handed over concurrently when the aligner can take it
        # (align_batch.concurrent = how many calls it accepts at a time; each call still gets ONE option set)
        groups = list(by_opts.items())
        width = int(getattr(align_batch, "concurrent", 1))
        if width > 2 and level < last_level:
            # Calls nothing waits for -- the ingroup pairs of the nodes: only the final files take their output -- are jobs of their own
            # that need not hold up the next level: they are handed over now and collected at the end of the phase, while the chains
            # (an ingroup against its outgroups, one after the other) go on.  Toil does the same with its independent jobs.
            held = []
            for opts, idx in groups:
                free_idx = [i for i in idx if calls[i].chain is None]
                rest = [i for i in idx if calls[i].chain is not None]
                if free_idx and rest:
                    free_jobs.append((opts, free_idx, pool.submit(gated, [(genomes[calls[i].target], query_fa[i]) for i in free_idx], opts, True)))
                    held.append((opts, rest))
                elif free_idx and any(calls[i].chain is not None for _, ix in groups for i in ix):
                    free_jobs.append((opts, free_idx, pool.submit(gated, [(genomes[calls[i].target], query_fa[i]) for i in free_idx], opts, True)))
                else:
                    held.append((opts, idx))
            groups = held
        split = int(getattr(align_batch, "split_above", 0))
        if width > 1 and split > 0:
            # a large group is handed over in two halves (two concurrent batched calls: their host-side work -- relay planting,
            # traceback, merge -- runs on two threads and the latency-bound tail launches of one overlap the other's work)
            halves = []
            for opts, idx in groups:
                if len(idx) >= split:
                    halves += [(opts, idx[0::2]), (opts, idx[1::2])]
                else:
                    halves.append((opts, idx))
            groups = halves
        if width > 1 and len(groups) > 1:
            # never more than `width` calls in flight, however many option sets (and halves) a level has: the aligner owns that
            # many contexts and no more
            results = list(pool.map(lambda g: gated([(genomes[calls[i].target], query_fa[i]) for i in g[1]], g[0]), groups))
        else:
            # (through the gate as well: jobs nothing waits for may be out on the aligner's other contexts)
            results = [gated([(genomes[calls[i].target], query_fa[i]) for i in idx], opts) for opts, idx in groups]
        for (opts, idx), outs in zip(groups, results):
            take(opts, idx, outs)

    try:
        for level in range(last_level + 1):
            run_level(level)
    except BaseException:
        for _, _, fut in free_jobs:                   # a level raised: nothing of this run stays in flight on the aligner's contexts
            fut.cancel()
        for _, _, fut in free_jobs:
            try:
                fut.result()
            except BaseException:                     # noqa: BLE001
                pass
        raise
    for opts, idx, fut in free_jobs:
        take(opts, idx, fut.result())
    # assemble: outgroup chains are merged innermost first (each level's sub-sequence coordinates fixed by dechunk --query,
    # make_ingroup_to_outgroup_alignments_3), then inverted so that the ingroup is the target
    parts: Dict[str, Dict[str, List[bytes]]] = {}     # (pieces first, ONE join per file: the files of a phase are megabytes)
    for i, c in enumerate(calls):
        node = parts.setdefault(c.node, {"ingroup": [], "outgroup": []})
        if c.kind == "ingroup":
            node["ingroup"].append(raw[i])
    chains: Dict[Tuple[str, str], List[int]] = {}
    for i, c in enumerate(calls):
        if c.chain is not None:
            chains.setdefault(c.chain, []).append(i)
    # make_ingroup_to_outgroup_alignments_3 merges a chain innermost first -- merged = raw[k] + dechunk(merged of k+1 ..) -- and
    # inverts the lot: per record that is invert(dechunk^k(record of level k)), in level order, which is what chain_share made
    for key, idx in chains.items():
        for i in sorted(idx, key=lambda i: calls[i].level):
            if i in finished:
                parts[key[0]]["outgroup"].append(finished[i].result())
    return {node: {kind: b"".join(p) for kind, p in files.items()} for node, files in parts.items()}
