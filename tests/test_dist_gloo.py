"""CPU, world_size 2 over gloo: the N>1 path of batched matching -- ssvio_amd.dist.window_matches (config 4: pair-blocks
round-robin, gather of DMatch lists) and ssvio_amd.dist.loop_candidates (config 5: super-block walk of the triangular pair
matrix, per-pair summary records, one all_gather).  These are the functions bench.py and an 8-GPU run execute; only the
per-rank compute object differs: there is no GPU in this container and the library has no CPU path, so a stand-in with
DeviceBlocks' three methods plays the rank's GPU with the oracle."""
import contextlib
import os
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NSETS, ROWS = 11, 96


def _sets():
    from ssvio_amd import synth
    sets = [synth.descriptor_sets(100 + s, ROWS, ROWS)[0] for s in range(NSETS)]
    sets[3] = sets[3][:50]                                   # ragged sets
    sets[7] = sets[7][:1]
    sets[9] = sets[9][:0]
    return sets


class OracleBlocks:
    """stand-in for ssvio_amd.dist.DeviceBlocks on a CPU: same three methods, results from the oracle"""
    device = torch.device("cpu")

    def __init__(self, sets, max_rows):
        self.sets, self.max_rows, self.calls = sets, max_rows, []

    def stream(self):
        return contextlib.nullcontext()

    def publish(self):
        pass

    def match(self, pairs, mode):
        from oracle import oracle as O
        fn = [O.match_bruteforce, lambda a, b: O.match_knn2_ratio(a, b, 50, 0.6), O.match_crosscheck][mode]
        self.pairs = np.asarray(pairs)
        self.res = [fn(self.sets[a], self.sets[b]) for a, b in self.pairs]
        self.calls.append(len(pairs))

    def summaries_into(self, dst, id_mode, id_base=0):
        from ssvio_amd import dist as D, _lib as L
        rec = np.zeros(len(self.res), L.REC_DTYPE)
        rec["pair"] = D.tri_id(self.pairs[:, 0], self.pairs[:, 1]) if id_mode == L.PAIR_ID_TRIANGULAR else id_base + np.arange(len(self.res))
        rec["n_matches"] = [len(r) for r in self.res]
        rec["sum_dist"] = [int(r["distance"].sum()) for r in self.res]
        dst.copy_(torch.from_numpy(rec.view(np.int32).reshape(-1, 4)))

    def lists_into(self, dst_m, dst_n):
        for k, r in enumerate(self.res):
            dst_m[k, :len(r)] = torch.from_numpy(r.view(np.int32).reshape(-1, 4))
            dst_n[k] = len(r)


def _expected():
    from ssvio_amd import dist as D, _lib as L
    from oracle import oracle as O
    sets = _sets()
    pairs = D.all_pairs(NSETS)
    rec = np.zeros(len(pairs), L.REC_DTYPE)
    for k, (a, b) in enumerate(pairs):
        r = O.match_bruteforce(sets[a], sets[b])
        rec[k] = (k, len(r), int(r["distance"].sum()))
    wpairs = np.array([[0, k] for k in range(1, NSETS)] + [[5, 5]], np.int32)
    lists = [O.match_crosscheck(sets[a], sets[b]) for a, b in wpairs]
    return pairs, rec, wpairs, lists


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ssvio_amd import dist as D, _lib as L
    blocks = OracleBlocks(_sets(), ROWS)
    rec = D.records_to_numpy(D.loop_candidates(blocks, NSETS, rank, world, sb=4))
    calls = list(blocks.calls)
    wpairs = np.array([[0, k] for k in range(1, NSETS)] + [[5, 5]], np.int32)
    work = {}
    for _ in range(2):                                        # twice: the staging buffers are reused
        lists, counts = D.lists_to_numpy(*D.window_matches(blocks, wpairs, rank, world, L.MODE_CROSSCHECK, work=work))
    q.put((rank, rec.tobytes(), calls, lists.view(np.uint8).tobytes(), counts.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_partitioning_covers_every_pair_once():
    from ssvio_amd import dist as D
    for nsets, sb, world in ((11, 4, 2), (24, 8, 8), (64, 8, 8), (5, 32, 3), (2, 1, 2), (4096, 64, 8)):
        plan, load = D.deal_super_blocks(nsets, sb, world)
        assert sum(load) == nsets * (nsets - 1) // 2
        assert max(load) - min(load) <= sb * sb                     # within one tile of each other
        if nsets <= 64:
            ids = np.concatenate([D.tri_id(*D.super_block_pairs(bi, bj, sb, nsets).T) for r in plan for bi, bj in r] + [np.zeros(0, np.int64)])
            assert sorted(ids.tolist()) == list(range(nsets * (nsets - 1) // 2))
    p = D.all_pairs(6)
    assert (p[:, 0] > p[:, 1]).all() and D.tri_id(p[:, 0], p[:, 1]).tolist() == list(range(15))
    assert sorted(D.shard_pairs(21, 0, 2).tolist() + D.shard_pairs(21, 1, 2).tolist()) == list(range(21))


def test_single_process_without_a_process_group():
    from ssvio_amd import dist as D, _lib as L
    pairs, want_rec, wpairs, want_lists = _expected()
    blocks = OracleBlocks(_sets(), ROWS)
    rec = D.records_to_numpy(D.loop_candidates(blocks, NSETS, 0, 1, sb=4))
    assert rec.tobytes() == want_rec.tobytes()
    assert max(blocks.calls) <= 16                                  # matcher scratch bounded by one super-block
    lists, counts = D.lists_to_numpy(*D.window_matches(blocks, wpairs, 0, 1, L.MODE_CROSSCHECK))
    assert counts.tolist() == [len(r) for r in want_lists]
    assert all(np.array_equal(lists[k, :len(r)].view(np.uint8), r.view(np.uint8)) for k, r in enumerate(want_lists))


def test_two_ranks_over_gloo_equal_the_single_process_result():
    sys.path.insert(0, ROOT)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs: p.join(60)
    assert all(p.exitcode == 0 for p in procs)
    pairs, want_rec, wpairs, want_lists = _expected()
    want = np.zeros((len(wpairs), ROWS), dtype=[("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])
    for k, r in enumerate(want_lists):
        want[k, :len(r)] = r
    ncalls = 0
    for rank, recbytes, calls, mbytes, cnt in res:
        assert recbytes == want_rec.tobytes()                       # every rank holds every pair's record, in pair order
        assert cnt == [len(r) for r in want_lists] and mbytes == want.view(np.uint8).tobytes()
        assert max(calls) <= 16
        ncalls += sum(calls)
    assert ncalls == len(pairs)                                     # the two ranks split the pair-blocks between them
