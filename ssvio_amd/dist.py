"""Multi-GPU batched matching (BASELINE configs 4 and 5).

Keyframe-pair blocks are independent units: they are dealt to the ranks with no exchange during compute, the descriptor
sets are replicated on every rank (4096 keyframes x 8000 x 32 B = 1.05 GB against 288 GB of HBM; only the work is
sharded), and ONE all_gather of fixed-size records at the end is the only collective (RCCL over xGMI when the process
group's backend is "nccl").  Everything between the matcher and the collective stays on the device.

  window_matches   config 4: a list of pair-blocks dealt round-robin, gather of the full DMatch lists
  loop_candidates  config 5: all unordered keyframe pairs, walked in super-blocks of the triangular block matrix with
                   bounded matcher scratch, every pair reduced on the device to a 16-byte {pair, n_matches, sum_dist}
                   record (ssvio_match_batch_summary_dev), gather of the records

The per-rank compute sits behind a small "blocks" object (stream / match / summaries_into / lists_into / publish).  DeviceBlocks is the product:
FeatureMatching on the rank's GPU.  tests/test_dist_gloo.py drives these same functions over gloo on the CPU with a
stand-in blocks object, because this library has no CPU path of its own.
"""

import numpy as np
import torch
import torch.distributed as dist

from ._lib import (DM_DTYPE, REC_DTYPE, MODE_BRUTEFORCE, MODE_CROSSCHECK, PAIR_ID_SEQUENTIAL, PAIR_ID_TRIANGULAR)

# ---------------------------------------------------------------- partitioning (pure host logic)
def shard_pairs(npairs, rank, world):
    """pair-block p goes to rank p % world (round-robin)"""
    return np.arange(rank, npairs, world, dtype=np.int64)

def all_pairs(nsets):
    """unordered keyframe pairs (query = later keyframe i, train = earlier keyframe j < i) in triangular order: row
    i * (i - 1) / 2 + j, the numbering SSVIO_PAIR_ID_TRIANGULAR gives the records"""
    j, i = np.triu_indices(nsets, 1)
    order = np.lexsort((j, i))
    return np.stack([i[order], j[order]], 1).astype(np.int32)

def tri_id(q, t):
    hi = np.maximum(q, t).astype(np.int64); lo = np.minimum(q, t).astype(np.int64)
    return hi * (hi - 1) // 2 + lo

def super_block_pairs(bi, bj, sb, nsets):
    """the pair-blocks (i, j), j < i, of tile (bi, bj) of the keyframe x keyframe block matrix (tiles of sb x sb keyframes)"""
    i = np.arange(bi * sb, min((bi + 1) * sb, nsets)); j = np.arange(bj * sb, min((bj + 1) * sb, nsets))
    ii, jj = np.meshgrid(i, j, indexing="ij")
    keep = jj < ii
    return np.stack([ii[keep], jj[keep]], 1).astype(np.int32)

def deal_super_blocks(nsets, sb, world):
    """Tiles (bi, bj), bj <= bi, of the lower triangle, dealt to `world` ranks: largest first, each to the rank with the
    least pair-blocks so far (ties: lowest rank) -- deterministic, so every rank computes the same deal.  Full tiles hold
    sb*sb pair-blocks, diagonal ones sb*(sb-1)/2.  -> (tiles per rank, pair-blocks per rank)"""
    nb = (nsets + sb - 1) // sb
    tiles = []
    for bi in range(nb):
        for bj in range(bi + 1):
            ni = min((bi + 1) * sb, nsets) - bi * sb; nj = min((bj + 1) * sb, nsets) - bj * sb
            n = ni * (ni - 1) // 2 if bi == bj else ni * nj
            if n:
                tiles.append((n, bi, bj))
    tiles.sort(key=lambda t: (-t[0], t[1], t[2]))
    plan = [[] for _ in range(world)]; load = [0] * world
    for n, bi, bj in tiles:
        r = min(range(world), key=lambda k: (load[k], k))
        plan[r].append((bi, bj)); load[r] += n
    return plan, load

# ---------------------------------------------------------------- the rank's GPU
class _DevBytes:
    """a device buffer owned by the library, presented to torch through __cuda_array_interface__ (no copy)"""
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}

class DeviceBlocks:
    """This rank's share of the work on its GPU: descriptor sets resident in HBM (set s = rows [s*set_stride_rows, +counts[s])
    of d_desc), matched by ssvio_match_batch_dev.  Matcher kernels, the device-to-device staging copies and the collective are
    ordered on one torch stream; nothing returns to the host."""

    def __init__(self, fm, d_desc, set_stride_rows, max_rows, d_counts, device):
        self.fm, self.d_desc, self.d_counts = fm, d_desc, d_counts
        self.set_stride_rows, self.max_rows = int(set_stride_rows), int(max_rows)
        self.device = torch.device(device)
        self.st = torch.cuda.Stream(device=self.device)
        fm.set_stream(self.st.cuda_stream)
        self._last = None

    def stream(self):
        return torch.cuda.stream(self.st)

    def match(self, pairs, mode):
        dm, dn = self.fm.match_batch_dev(self.d_desc, self.set_stride_rows, self.max_rows, self.d_counts, pairs, mode)
        self._last = (dm, dn, len(pairs))

    def summaries_into(self, dst, id_mode, id_base=0):
        """dst: int32 [n_last_pairs, 4] slice of a device tensor; written by k_pair_summary on the matcher's stream"""
        assert dst.is_cuda and dst.dtype == torch.int32 and dst.is_contiguous() and dst.shape == (self._last[2], 4)
        self.fm.batch_summary_dev(dst, id_mode, id_base)

    def lists_into(self, dst_m, dst_n):
        """dst_m: int32 [n, max_rows, 4], dst_n: int32 [n] device slices <- the matcher's output buffers (device-to-device)"""
        dm, dn, n = self._last
        src_m = torch.as_tensor(_DevBytes(dm, n * self.max_rows * 16), device=self.device).view(torch.int32)
        src_n = torch.as_tensor(_DevBytes(dn, n * 4), device=self.device).view(torch.int32)
        dst_m.view(-1).copy_(src_m); dst_n.copy_(src_n)

    def loop_walk(self, nsets, sb, rank, world, mode, local):
        """this rank's super-blocks in one native call (ssvio_loop_candidates_dev): records into local[:load[rank]]"""
        assert local.is_cuda and local.dtype == torch.int32 and local.is_contiguous() and local.shape[1] == 4
        return self.fm.loop_candidates_dev(self.d_desc, self.set_stride_rows, self.max_rows, self.d_counts, nsets, sb, rank, world, mode,
                                           local, local.shape[0])

    def publish(self):
        """results produced on the blocks' stream become visible to the caller's current torch stream"""
        torch.cuda.current_stream(self.device).wait_stream(self.st)

    def sync(self):
        self.st.synchronize()

# ---------------------------------------------------------------- the collective
def _backend(group=None):
    return dist.get_backend(group) if dist.is_available() and dist.is_initialized() else None

def gather_fixed(local, world, group=None):
    """all_gather of one fixed-size tensor per rank -> [world, *local.shape] on local's device.  world == 1 without a
    process group is a view.  A non-NCCL backend (the gloo rehearsal) with device tensors goes through host tensors."""
    if world == 1 and _backend(group) is None:
        return local.unsqueeze(0)
    out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    if local.is_cuda and _backend(group) != "nccl":
        torch.cuda.current_stream(local.device).synchronize()
        parts = [torch.empty(local.shape, dtype=local.dtype) for _ in range(world)]
        dist.all_gather(parts, local.cpu(), group=group)
        out.copy_(torch.stack(parts))
    else:
        dist.all_gather(list(out.unbind(0)), local, group=group)
    return out

def _hand_over(blocks, *tensors):
    """results were allocated while the blocks' stream was current: tell the caching allocator that the caller's stream reads
    them too, so their memory is not handed out again (to work on the blocks' stream) while that stream still does"""
    for t in tensors:
        if t.is_cuda:
            t.record_stream(torch.cuda.current_stream(t.device))

# ---------------------------------------------------------------- config 4
def window_matches(blocks, pairs, rank, world, mode=MODE_CROSSCHECK, group=None, work=None):
    """pairs[npairs][2] (query set, train set) dealt round-robin; every rank matches its share and ONE all_gather returns all
    lists.  -> (lists int32 [npairs, max_rows, 4] = DMatch rows, counts int32 [npairs]) on blocks.device, in pair order.
    `work`: a dict that keeps the staging buffers between calls."""
    pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
    npairs, R = len(pairs), blocks.max_rows
    mine = shard_pairs(npairs, rank, world)
    per = (npairs + world - 1) // world
    work = work if work is not None else {}
    key = ("window", per, R)
    with blocks.stream():
        if key not in work:
            work[key] = torch.zeros(per * R * 4 + per, dtype=torch.int32, device=blocks.device)
        local = work[key]
        lists = local[:per * R * 4].view(per, R, 4); counts = local[per * R * 4:]
        counts.fill_(-1)
        if len(mine):
            blocks.match(pairs[mine], mode)
            blocks.lists_into(lists[:len(mine)], counts[:len(mine)])
        g = gather_fixed(local, world, group)                                   # [world, per*R*4 + per]
        gl = g[:, :per * R * 4].reshape(world, per, R, 4); gc = g[:, per * R * 4:]
        # pair p sits on rank p % world in slot p // world
        p = torch.arange(npairs, device=blocks.device)
        res = gl[p % world, p // world], gc[p % world, p // world]
    blocks.publish()
    _hand_over(blocks, *res)
    return res

def lists_to_numpy(lists, counts):
    m = np.ascontiguousarray(lists.cpu().numpy()).view(DM_DTYPE).reshape(lists.shape[0], lists.shape[1])
    return m, counts.cpu().numpy()

# ---------------------------------------------------------------- config 5
def loop_candidates(blocks, nsets, rank, world, sb=32, mode=MODE_BRUTEFORCE, group=None, work=None):
    """All unordered keyframe pairs of `nsets` descriptor sets (query = later keyframe, MatchByBruteForce per pair-block,
    src/FeatureMatching.cpp:20-55).  The triangular block matrix is tiled into super-blocks of sb x sb keyframes
    (deal_super_blocks); a rank walks its tiles one ssvio_match_batch_dev call each, so the matcher's scratch is bounded by
    sb*sb pair-blocks, and reduces every pair on the device to a 16-byte record {triangular pair id, n_matches, sum_dist}.
    One all_gather of the records at the end.  -> int32 tensor [npairs, 4] on blocks.device, row = pair id
    (records_to_numpy gives the REC_DTYPE view)."""
    plan, load = deal_super_blocks(nsets, sb, world)
    per = max(max(load), 1)
    npairs = nsets * (nsets - 1) // 2
    work = work if work is not None else {}
    key = ("loop", per)
    with blocks.stream():
        if key not in work:
            work[key] = torch.empty((per, 4), dtype=torch.int32, device=blocks.device)
        local = work[key]
        local[load[rank]:, 0] = -1                                               # padding records
        if hasattr(blocks, "loop_walk"):
            # the product path: the whole walk of this rank is one C-ABI call (same deal, computed natively; checked against this
            # module's deal_super_blocks in tests/test_cabi_and_host.py)
            o = blocks.loop_walk(nsets, sb, rank, world, mode, local)
        else:
            o = 0
            for bi, bj in plan[rank]:
                pairs = super_block_pairs(bi, bj, sb, nsets)
                blocks.match(pairs, mode)
                blocks.summaries_into(local[o:o + len(pairs)], PAIR_ID_TRIANGULAR)
                o += len(pairs)
        assert o == load[rank]
        g = gather_fixed(local, world, group).reshape(world * per, 4)
        ids = g[:, 0].long()
        keep = ids >= 0
        out = torch.zeros((npairs, 4), dtype=torch.int32, device=blocks.device)
        out[ids[keep]] = g[keep]
    blocks.publish()
    _hand_over(blocks, out)
    return out

def records_to_numpy(rec):
    return np.ascontiguousarray(rec.cpu().numpy()).view(REC_DTYPE).reshape(-1)
