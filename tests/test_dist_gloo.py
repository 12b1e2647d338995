"""CPU, world_size 2 over gloo: the N>1 path of batched matching -- pair-block sharding and the
gather of fixed-size records.  The per-rank compute is played by the oracle here (no GPU in this
container); on the GPU box the same functions are driven by FeatureMatching.match_batch_dev."""
import os
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ssvio_amd import dist as D, synth, _lib
    from oracle import oracle as O
    nsets, rows = 7, 96
    sets = [synth.descriptor_sets(100 + s, rows, rows)[0] for s in range(nsets)]
    sets[3] = sets[3][:50]                                   # ragged set
    pairs = D.all_pairs(nsets)

    def match_fn(local_pairs):
        m = np.zeros((len(local_pairs), rows), _lib.DM_DTYPE); c = np.zeros(len(local_pairs), np.int32)
        for k, (a, b) in enumerate(local_pairs):
            r = O.match_crosscheck(sets[a], sets[b])
            c[k] = len(r); m[k, :len(r)] = r
        return m, c

    allm, cnt = D.match_pairs_sharded(match_fn, pairs, rows, rank, world, torch.device("cpu"), full_lists=True)
    rec = D.match_pairs_sharded(match_fn, pairs, rows, rank, world, torch.device("cpu"), full_lists=False)
    q.put((rank, allm.view(np.uint8).tobytes(), cnt.tolist(), rec.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_matching_two_ranks_equals_single_process():
    sys.path.insert(0, ROOT)
    from ssvio_amd import dist as D, synth, _lib
    from oracle import oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs: p.join(60)
    assert all(p.exitcode == 0 for p in procs)
    # reference: every pair in one process
    nsets, rows = 7, 96
    sets = [synth.descriptor_sets(100 + s, rows, rows)[0] for s in range(nsets)]
    sets[3] = sets[3][:50]
    pairs = D.all_pairs(nsets)
    assert len(pairs) == 21 and sorted(D.shard_pairs(21, 0, 2).tolist() + D.shard_pairs(21, 1, 2).tolist()) == list(range(21))
    want_cnt = []; want = np.zeros((len(pairs), rows), _lib.DM_DTYPE)
    for k, (a, b) in enumerate(pairs):
        r = O.match_crosscheck(sets[a], sets[b]); want_cnt.append(len(r)); want[k, :len(r)] = r
    for rank, mbytes, cnt, recbytes in res:
        assert cnt == want_cnt
        assert mbytes == want.view(np.uint8).tobytes()
        rec = np.frombuffer(recbytes, D.REC_DTYPE)
        assert rec["pair"].tolist() == list(range(21)) and rec["n_matches"].tolist() == want_cnt
        assert rec["sum_dist"].tolist() == [int(want[k, :c]["distance"].sum()) for k, c in enumerate(want_cnt)]
