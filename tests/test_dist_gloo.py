"""world_size-2 gloo test of the N>1 path on CPU: chunk-pair sharding + the gather of the final PAF
bytes must give the same bytes as a single process, in chunk-pair order."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cactus_amd.multigpu import assign_pairs, blast_pairs_sharded


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
