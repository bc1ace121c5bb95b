"""world_size-2 gloo test of the N>1 path on CPU: chunk-pair sharding + the gather of the final PAF
bytes must give the same bytes as a single process, in chunk-pair order."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cactus_amd.multigpu import assign_pairs, assign_target_major, blast_pairs_sharded, chain_parts_sharded, gather_bytes


def _fake_align(pair):
    i, a, b = pair
    return ("pair%d\t%d\t%d\n" % (i, a, b) * (i % 3)).encode()       # some pairs produce nothing


def _worker(rank, world, port, pairs, weights, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = blast_pairs_sharded(pairs, weights, _fake_align, dist, rank, world, torch.device("cpu"))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_assign_pairs_lpt_is_deterministic_and_balanced():
    w = [9, 1, 8, 2, 7, 3, 6, 4, 5]
    a = assign_pairs(w, 4)
    assert sorted(sum(a, [])) == list(range(9)) and a == assign_pairs(w, 4)
    loads = [sum(w[i] for i in part) for part in a]
    assert max(loads) - min(loads) <= 3
    assert assign_pairs([1.0] * 3, 8)[:3] == [[0], [1], [2]]


def test_target_major_ownership_builds_a_table_once_and_never_more_than_the_bound_per_rank():
    """rank g owns the target chunks i mod N with all their units; with fewer target chunks than ranks a chunk's column of
    units is shared by a group of ranks.  Every unit exactly once, no rank with more than ceil(Na / N) target chunks, a chunk on ONE
    rank whenever Na >= N, nobody idle while a column has units to spare -- for the shapes of both chunk-scale legs (chr20: 3 x 3
    chunk pairs, also as strand halves; the human-mouse stand-in: 7 x 6) at 1, 2, 3 and 8 ranks."""
    for na, nb, halves in ((3, 3, False), (3, 3, True), (7, 6, False), (104, 91, False)):
        units = [(i, j, h) for i in range(na) for j in range(nb) for h in ((1, 2) if halves else (0,))]
        targets = [u[0] for u in units]
        weights = [(30.0 + i) * (30.0 + (j * 7) % 5) * (0.5 if h else 1.0) for i, j, h in units]
        for world in (1, 2, 3, 8):
            shares = assign_target_major(targets, weights, world)
            assert shares == assign_target_major(targets, weights, world)
            assert sorted(sum(shares, [])) == list(range(len(units)))
            per_rank = [{targets[u] for u in sh} for sh in shares]
            assert max(len(t) for t in per_rank) <= -(-na // world)
            if na >= world:
                owners = {t: [r for r in range(world) if t in per_rank[r]] for t in range(na)}
                assert all(len(o) == 1 and o[0] == t % world for t, o in owners.items())
                assert sum(len(t) for t in per_rank) == na                      # every table built exactly once
            else:
                assert all(sh for sh in shares[:min(world, len(units))])          # every rank has work
                loads = [sum(weights[u] for u in sh) for sh in shares if sh]
                assert max(loads) <= 2.5 * (sum(loads) / len(loads))


def test_two_rank_gloo_gather_equals_single_process():
    pairs = [(i, 1000 + i, 2000 - i) for i in range(7)]
    weights = [float(a * b) for _, a, b in pairs]
    single = blast_pairs_sharded(pairs, weights, _fake_align, None, 0, 1, torch.device("cpu"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, pairs, weights, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got[1] is None
    assert got[0] == single == b"".join(_fake_align(p) for p in pairs)


def _big_gather_worker(rank, world, port, sizes, q):
    import hashlib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = sizes[rank]
    # (a payload that is cheap to make and to check: the rank's byte, a counter every 4 KiB)
    buf = bytearray([65 + rank]) * n
    for k in range(0, n, 4096):
        buf[k:k + 8] = k.to_bytes(8, "little")
    mine = hashlib.md5(buf).hexdigest()
    got = gather_bytes(bytes(buf), dist, rank, world, torch.device("cpu"))
    del buf
    if rank == 0:
        q.put(("gathered", [(len(x), hashlib.md5(x).hexdigest()) for x in got]))
    q.put((rank, (n, mine)))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_of_payloads_of_very_different_and_large_sizes():
    """The one exchange of the multi-GPU path at the scale SURVEY 8e names for a whole-genome run ("~2-5 GB total"): three ranks whose PAF
    payloads are 600 MB, nothing at all and 150 MB.  gather_bytes sends every payload point-to-point into a buffer of exactly its size --
    nothing padded to the largest, nothing of size world x max allocated -- and rank 0 gets every byte (md5 per payload)."""
    sizes = [150 << 20, 0, 600 << 20] if os.environ.get("MIBLAST_TEST_BIG_GATHER", "1") != "0" else [3 << 20, 0, 12 << 20]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_big_gather_worker, args=(r, 3, port, sizes, q)) for r in range(3)]
    for p in procs:
        p.start()
    msgs = dict(q.get(timeout=600) for _ in range(4))
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    assert msgs["gathered"] == [msgs[r] for r in range(3)]
    assert [n for n, _ in msgs["gathered"]] == sizes


# ---- chaining stage: per-contig parts sharded over ranks (the job itself is the oracle here: CPU test) ------------------------------
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_chain_job(part: bytes) -> bytes:
    import subprocess
    o = os.path.join(_ROOT, "oracle", "oracle_paffy")
    chain = f"{o} chain --maxGapLength 1000000 --chainGapOpen 5000 --chainGapExtend 1 --trimFraction 1.0"
    cmd = f"{chain} | {o} tile | {o} trim --trimIdentity 0.2 | {o} filter --maxTileLevel 1 | {chain} | {o} filter --minChainScore 10000"
    return subprocess.run(["bash", "-o", "pipefail", "-c", cmd], input=part, capture_output=True, check=True).stdout


def _chain_worker(rank, world, port, parts, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = chain_parts_sharded(parts, _oracle_chain_job, dist, rank, world, torch.device("cpu"))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_chaining_of_contig_parts_equals_single_process():
    from cactus_amd import gen
    parts = [gen.random_paf(70 + k, n_series=3 + k, noise=5 * k, n_q=1, n_t=2).replace("id=Q|chr0", f"id=Q|chr{k}").encode() for k in range(5)]
    single = chain_parts_sharded(parts, _oracle_chain_job, None, 0, 1, torch.device("cpu"))
    assert single == b"".join(_oracle_chain_job(p) for p in parts) and single
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_chain_worker, args=(r, 2, port, parts, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got[1] is None and got[0] == single


def _reduce_worker(rank, world, port, q):
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    argv, sys.argv = sys.argv, ["bench.py"]
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sys.argv = argv
    # what a rank hands over after its timed steps: counters (here: different on every rank) and its barrier-to-barrier time
    tot = {"dp_cells": 1000.0 * (rank + 1), "seed_hits": 7.0, "t_dp_kernel_ms": 0.5 + rank}
    elapsed, summed = bench.reduce_totals(tot, 0.010 * (rank + 1), dist, torch.device("cpu"))
    q.put((rank, elapsed, summed))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_reduction_of_the_bench_line():
    """bench.py's reduce_totals at world_size 2 (the N > 1 line: value = the units of all ranks / the slowest rank's time): counters are
    summed over the ranks, the time is the maximum, and every rank gets the same figures."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reduce_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    for _, elapsed, summed in got:
        assert abs(elapsed - 0.020) < 1e-12
        assert summed == {"dp_cells": 3000.0, "seed_hits": 14.0, "t_dp_kernel_ms": 2.0}
