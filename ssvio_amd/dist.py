"""Multi-GPU batched matching (BASELINE configs 4 and 5): keyframe-pair blocks are independent
units, so they are dealt to ranks with no exchange during compute; one all_gather of fixed-size
per-pair records at the end is the only collective (RCCL over xGMI when the backend is "nccl").

The descriptor sets are replicated on every rank (4096 keyframes x 8000 x 32 B = 1.05 GB against
288 GB of HBM), only the work is sharded.
"""
import numpy as np
import torch
import torch.distributed as dist

REC_DTYPE = np.dtype([("pair", "<i4"), ("n_matches", "<i4"), ("sum_dist", "<i8")])   # 16-B per-pair summary record


def shard_pairs(npairs, rank, world):
    """pair-block p goes to rank p % world (round-robin: triangular all-pairs lists stay balanced)"""
    return np.arange(rank, npairs, world, dtype=np.int64)


def all_pairs(nsets):
    """unordered keyframe pairs (i < j), the loop-candidate workload of config 5"""
    i, j = np.triu_indices(nsets, 1)
    return np.stack([j, i], 1).astype(np.int32)          # (query = later keyframe, train = earlier)


def summarize(pair_ids, matches, counts):
    """per-pair fixed-size record {pairId, nGood, sum of distances}"""
    rec = np.zeros(len(pair_ids), REC_DTYPE)
    rec["pair"] = pair_ids
    rec["n_matches"] = counts
    if matches is not None:
        for k, c in enumerate(counts):
            rec["sum_dist"][k] = int(matches[k, :c]["distance"].sum())
    return rec


def gather_records(rec, npairs, world, device):
    """all_gather of per-pair records (fixed size: every rank pads its shard to ceil(npairs/world))"""
    per = (npairs + world - 1) // world
    buf = np.zeros(per, REC_DTYPE); buf["pair"] = -1
    buf[:len(rec)] = rec
    t = torch.from_numpy(buf.view(np.int32).reshape(per, 4).copy()).to(device)
    if world > 1:
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        allrec = torch.cat(out).cpu().numpy()
    else:
        allrec = t.cpu().numpy()
    allrec = np.ascontiguousarray(allrec).view(REC_DTYPE).reshape(-1)
    allrec = allrec[allrec["pair"] >= 0]
    return allrec[np.argsort(allrec["pair"], kind="stable")]


def gather_matches(matches, counts, pair_ids, npairs, max_rows, world, device):
    """all_gather of the full best-match lists: [npairs][max_rows] DMatch rows + counts (config 4:
    64 x 4000 x 16 B = 4 MB in total)"""
    per = (npairs + world - 1) // world
    m = np.zeros((per, max_rows, 4), np.int32); c = np.full((per, 2), -1, np.int32)
    m[:len(pair_ids)] = matches.view(np.int32).reshape(len(pair_ids), max_rows, 4)
    c[:len(pair_ids), 0] = pair_ids; c[:len(pair_ids), 1] = counts
    tm = torch.from_numpy(m).to(device); tc = torch.from_numpy(c).to(device)
    if world > 1:
        om = [torch.empty_like(tm) for _ in range(world)]; oc = [torch.empty_like(tc) for _ in range(world)]
        dist.all_gather(om, tm); dist.all_gather(oc, tc)
        tm = torch.cat(om); tc = torch.cat(oc)
    tm = tm.cpu().numpy(); tc = tc.cpu().numpy()
    keep = tc[:, 0] >= 0
    order = np.argsort(tc[keep, 0], kind="stable")
    from ._lib import DM_DTYPE
    allm = np.ascontiguousarray(tm[keep][order]).view(DM_DTYPE).reshape(-1, max_rows)
    return allm, tc[keep][order][:, 1]


def match_pairs_sharded(match_fn, pairs, max_rows, rank, world, device, full_lists=True):
    """match_fn(local_pairs) -> (matches[n_local, max_rows] DM_DTYPE, counts[n_local]) runs this rank's
    share on its GPU (FeatureMatching.match_batch_dev + fetch); the results of all ranks come back
    on every rank in pair order."""
    mine = shard_pairs(len(pairs), rank, world)
    if len(mine):
        m, c = match_fn(pairs[mine])
    else:
        from ._lib import DM_DTYPE
        m, c = np.zeros((0, max_rows), DM_DTYPE), np.zeros(0, np.int32)
    if full_lists:
        return gather_matches(m, c, mine, len(pairs), max_rows, world, device)
    return gather_records(summarize(mine, m, c), len(pairs), world, device)
