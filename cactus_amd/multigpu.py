"""Multi-GPU driver of the blast phase: chunk pairs are independent (they are separate Toil jobs in the
reference, /root/reference/src/cactus/paf/local_alignment.py:395-405), so the path shards with no
data-path collective.  One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm,
"gloo" in the CPU tests); the single exchange is the gather of the final PAF bytes to rank 0
(SURVEY.md 8e).  Output is assembled in chunk-pair order, so it is byte-identical for any world size."""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence


def assign_pairs(weights: Sequence[float], world_size: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of chunk pairs (weight ~ len(A) * len(B)) to ranks.
    Deterministic: ties go to the lower pair index, then the lower rank."""
    order = sorted(range(len(weights)), key=lambda i: (-weights[i], i))
    load = [0.0] * world_size
    out: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += weights[i]
    for lst in out:
        lst.sort()
    return out


def assign_target_major(unit_targets: Sequence[int], weights: Sequence[float], world_size: int) -> List[List[int]]:
    """Target-major ownership of the work units of ONE genome pair (the jobs are the independent run_lastz jobs of
    /root/reference/src/cactus/paf/local_alignment.py:395-405): unit_targets[u] = the target chunk of unit u (a chunk pair, or a
    (chunk pair, query strand) half).  With at least as many target chunks as ranks, rank g owns the target chunks {i : i mod N == g}
    and every unit of an owned chunk -- a chunk's seed table is built once, on one GPU, and all its query chunks stream through it.
    With FEWER target chunks than ranks a chunk's column of units is shared: the ranks are split between the chunks in proportion to
    the columns' weights (every chunk at least one), the units of a column are dealt longest first to the column's ranks, and each of
    those ranks builds that one table.  Either way no rank builds more than ceil(Na / N) tables.  Deterministic (every rank
    computes the same deal); returns the unit indices per rank, ascending."""
    targets = sorted(set(unit_targets))
    na = len(targets)
    out: List[List[int]] = [[] for _ in range(world_size)]
    if na == 0:
        return out
    col = {t: [u for u in range(len(unit_targets)) if unit_targets[u] == t] for t in targets}
    if na >= world_size:
        for k, t in enumerate(targets):
            out[k % world_size] += col[t]
    else:
        colw = {t: sum(weights[u] for u in col[t]) for t in targets}
        share = {t: 1 for t in targets}
        for _ in range(world_size - na):                      # one more rank at a time to the column with the most weight per rank
            t = max(targets, key=lambda x: (colw[x] / share[x], -x))
            share[t] += 1
        first = 0
        for t in targets:
            ranks = list(range(first, first + share[t]))
            first += share[t]
            load = {r: 0.0 for r in ranks}
            for u in sorted(col[t], key=lambda i: (-weights[i], i)):
                r = min(ranks, key=lambda k: (load[k], k))
                out[r].append(u)
                load[r] += weights[u]
    for lst in out:
        lst.sort()
    return out


def share_host_cores(local_world_size: int) -> int:
    """One rank per GPU on a node: every rank takes an equal share of the host cores for its worker threads (the reference
    passes job.cores as --num_threads for the same reason, /root/reference/src/cactus/paf/local_alignment.py:58).
    Returns the threads now in use by this rank."""
    import os
    from cactus_amd import miblast
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return miblast.set_host_threads(max(1, min(16, cores // max(1, local_world_size))))


def gather_bytes(payload: bytes, dist, rank: int, world_size: int, device) -> Optional[List[bytes]]:
    """Variable-length gather to rank 0 (the 'gatherv' of SURVEY 8e): all_gather of the byte counts, then every peer sends its
    bytes straight to rank 0 (point-to-point: over xGMI every peer has a link of its own to GPU 0) into a tensor of exactly that
    size -- nothing is padded to the largest payload, nothing of size world x max is ever allocated (a human-mouse run gathers
    gigabytes)."""
    import torch
    if world_size == 1:
        return [payload]
    n = torch.tensor([len(payload)], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world_size)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    if rank != 0:
        if payload:
            dist.send(torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(device), dst=0)
        return None
    # all receives are posted before any is waited for: the peers' transfers overlap (each has a link of its own to GPU 0)
    bufs = {src: torch.empty(sizes[src], dtype=torch.uint8, device=device) for src in range(1, world_size) if sizes[src]}
    reqs = [dist.irecv(buf, src=src) for src, buf in bufs.items()]
    for r in reqs:
        r.wait()
    return [payload] + [bufs[src].cpu().numpy().tobytes() if src in bufs else b"" for src in range(1, world_size)]


def fasta_names(fasta: bytes):
    """names (first word of the header) of the records of FASTA text, in file order"""
    return [line[1:].split()[0] if line[1:].split() else b"" for line in fasta.splitlines() if line.startswith(b">")]


def merge_strand_pafs(plus: bytes, minus: bytes, query_names) -> bytes:
    """The PAF of a chunk pair from its two halves `--strand=plus` and `--strand=minus` (miblast_params.strands): lastz writes, query
    sequence by query sequence in FILE order, the '+' alignments and then the '-' ones, and a half holds its lines in
    the same order.  query_names: the query file's record names in order (fasta_names)."""
    def groups(paf):
        out = {}
        for line in paf.splitlines(True):
            out.setdefault(line.split(b"\t", 1)[0], []).append(line)
        return out
    gp, gm = groups(plus), groups(minus)
    known = set(query_names)
    assert set(gp) <= known and set(gm) <= known, "a PAF query name that is not in the query file"
    return b"".join(b"".join(gp.get(n, [])) + b"".join(gm.get(n, [])) for n in query_names)


def _frame(index: int, paf: bytes) -> bytes:
    return index.to_bytes(8, "little") + len(paf).to_bytes(8, "little") + paf


def _unframe(blob: bytes):
    pos = 0
    while pos < len(blob):
        idx = int.from_bytes(blob[pos:pos + 8], "little")
        n = int.from_bytes(blob[pos + 8:pos + 16], "little")
        yield idx, blob[pos + 16:pos + 16 + n]
        pos += 16 + n


def blast_pairs_sharded(pairs: Sequence, weights: Sequence[float], align_fn: Callable[[object], bytes], dist, rank: int,
                        world_size: int, device) -> Optional[bytes]:
    """Aligns pairs[i] for the indices this rank owns, gathers the framed PAFs to rank 0 and returns
    the concatenation in pair order there (None elsewhere)."""
    mine = assign_pairs(weights, world_size)[rank]
    blob = b"".join(_frame(i, align_fn(pairs[i])) for i in mine)
    gathered = gather_bytes(blob, dist, rank, world_size, device)
    if gathered is None:
        return None
    by_index = {}
    for part in gathered:
        for idx, paf in _unframe(part):
            by_index[idx] = paf
    assert sorted(by_index) == list(range(len(pairs)))
    return b"".join(by_index[i] for i in range(len(pairs)))


def chain_parts_sharded(parts: Sequence[bytes], job_fn: Callable[[bytes], bytes], dist, rank: int, world_size: int, device) -> Optional[bytes]:
    """The chaining stage over several GPUs: the per-contig parts chain_alignments makes with `paffy split_file`
    (/root/reference/src/cactus/paf/local_alignment.py:636-657) are independent jobs, exactly like chunk pairs, so they shard the
    same way -- longest part first to the least loaded rank, no data-path collective, one gather of the outputs to rank 0, which
    returns them concatenated in part order (what merge_processed_alignments does).  job_fn(part_text) -> output text is one
    chain_tile_trim_filter_one_contig job (mipaf.PafSet...chain_tile_trim_filter on that rank's GPU)."""
    return blast_pairs_sharded(parts, [float(len(p)) for p in parts], job_fn, dist, rank, world_size, device)


def align_pairs_concurrent(pairs, params, device: int = 0, workers: int = 4):
    """Several chunk pairs of ONE GPU in flight at once (the reference schedules many single-threaded lastz jobs per
    node, /root/reference/src/cactus/paf/local_alignment.py:399-405).  A single pair cannot fill an MI355X -- its gapped
    stage is a handful of long, row-sequential DPs -- so independent pairs are overlapped from `workers` host threads,
    each with its own miblast context (own HIP stream and workspace; the library is re-entrant per context and ctypes
    releases the GIL during calls).  pairs: list of (target_fasta_bytes, query_fasta_bytes); returns PAF bytes per pair,
    in input order."""
    from concurrent.futures import ThreadPoolExecutor
    import threading
    from cactus_amd import miblast
    local = threading.local()
    out = [None] * len(pairs)

    def work(i):
        if not hasattr(local, "ctx"):
            local.ctx = miblast.Context(device)
        tf, qf = pairs[i]
        T, Q = local.ctx.seqset_from_fasta_bytes(tf), local.ctx.seqset_from_fasta_bytes(qf)
        try:
            r = local.ctx.align(T, Q, params, details=False)
        finally:
            T.close(); Q.close()
        out[i] = (r.paf, r.stats)

    with ThreadPoolExecutor(max_workers=max(1, workers)) as ex:
        list(ex.map(work, range(len(pairs))))
    return out
