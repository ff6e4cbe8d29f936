"""Query-level sharding over the GPUs of one node and the single exchange at the end of a job.

Independent planning queries (map x seed, or the sub-queries of a TAMP search, reference
src/map_shelves_tamp_rrt.rs:163-291) need no communication while they grow: query q runs on rank
q mod world_size.  The one collective step lives behind the C ABI (porrt_exchange_best, csrc/porrt_exchange.hpp:
ncclAllGather of a 16-byte entry per map and rank, per map the first minimum of (cost, rank), ncclBroadcast of the
winner's node arrays device to device).  This module is the caller: the partition, and the rendezvous of the
communicator over whatever process group the host program has (here torch.distributed: "nccl" on the GPU node,
"gloo" in the CPU tests).
"""
import numpy as np


def queries_of_rank(n_queries, rank, world_size):
    """Round-robin partition: query q -> rank q mod world_size."""
    return list(range(rank, n_queries, world_size))


def make_comm(device, dist=None, _factory=None):
    """porrt_comm for this process: rank 0 makes the RCCL unique id, the process group carries its 128 bytes.
    Collective over `dist`: either every rank returns a communicator or every rank raises (a rank that cannot make the id or
    its communicator tells the others through the process group, so that none walks on into a collective alone).
    _factory: (unique_id, Comm) stand-ins for the CPU tests of this protocol."""
    from .engine import Comm
    make_id, make = _factory if _factory else (Comm.unique_id, Comm)
    if dist is None or not dist.is_initialized():
        return make(device, 0, 1, make_id())
    rank, world = dist.get_rank(), dist.get_world_size()
    uid, err = None, None
    if rank == 0:
        try:
            uid = make_id()
        except Exception as ex:              # noqa: BLE001
            err = "rank 0: %s" % ex
    box = [(uid, err)]
    dist.broadcast_object_list(box, src=0)
    uid, err = box[0]
    comm = None
    if uid is not None:
        try:
            comm = make(device, rank, world, uid)
        except Exception as ex:              # noqa: BLE001
            err = "rank %d: %s" % (rank, ex)
    said = [None] * world
    dist.all_gather_object(said, None if comm is not None else err)
    bad = [e for e in said if e]
    if bad:
        if comm is not None:
            comm.close()
        raise RuntimeError("make_comm: " + bad[0])
    return comm


def exchange_best_per_map(comm, engines, map_ids, n_maps):
    """The exchange: winners[m] = (cost, rank, n_nodes) of the best tree of map m over all ranks (the same on every rank);
    the trees themselves stay on the device, comm.tree(m) fetches one."""
    return comm.exchange_best(engines, map_ids, n_maps)


def decide_from_gathered(entries_by_rank):
    """Step 2 of the exchange alone (porrt_exchange_decide, host code): per map the first minimum of (cost, rank)."""
    from .engine import BEST_ENTRY, exchange_decide
    return exchange_decide(np.ascontiguousarray(entries_by_rank, dtype=BEST_ENTRY))


def agree_from_gathered(words_by_rank, my_rank):
    """Step 0 of the exchange alone (porrt_exchange_agree, host code): from every rank's (status, n_maps) word, what this rank
    returns -- 0 only if every rank is fine and all were called with the same number of maps."""
    from .engine import exchange_agree
    return exchange_agree(words_by_rank, my_rank)
