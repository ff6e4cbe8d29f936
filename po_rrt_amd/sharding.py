"""Query-level sharding over the GPUs of one node and the single exchange at the end of a job.

Independent planning queries (map x seed, or the sub-queries of a TAMP search, reference
src/map_shelves_tamp_rrt.rs:163-291) need no communication while they grow: query q runs on rank
q mod world_size.  The one collective step: all ranks learn every rank's best path cost
(all_gather of one f64), and the winning tree (xy f64 x2, parent i64, dist_root f64) is broadcast from
its owner.  Works with any torch.distributed backend ("nccl" = RCCL over xGMI on the GPU node, "gloo" on CPU).
"""
import numpy as np


def queries_of_rank(n_queries, rank, world_size):
    """Round-robin partition: query q -> rank q mod world_size."""
    return list(range(rank, n_queries, world_size))


def exchange_best_tree(my_cost, xy, parent, dist_root, dist=None, device="cpu"):
    """Returns (winner_rank, winner_cost, xy, parent, dist_root) of the globally best tree on every rank.

    `my_cost` is this rank's best path cost (inf = no solution, rrt.rs:192).  Ties go to the lowest rank.
    With dist=None (single process) the inputs are returned unchanged.
    """
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return 0, float(my_cost), xy, parent, dist_root
    world, rank = dist.get_world_size(), dist.get_rank()
    costs = [torch.zeros(1, dtype=torch.float64, device=device) for _ in range(world)]
    dist.all_gather(costs, torch.tensor([float(my_cost)], dtype=torch.float64, device=device))
    costs = [float(c.item()) for c in costs]
    winner = int(np.argmin(costs))                    # first minimum: lowest rank wins ties
    n = torch.tensor([len(parent)], dtype=torch.int64, device=device)
    dist.broadcast(n, src=winner)
    n = int(n.item())
    if rank == winner:
        t_xy = torch.from_numpy(np.ascontiguousarray(xy, dtype=np.float64)).to(device)
        t_par = torch.from_numpy(np.ascontiguousarray(parent, dtype=np.int64)).to(device)
        t_dist = torch.from_numpy(np.ascontiguousarray(dist_root, dtype=np.float64)).to(device)
    else:
        t_xy = torch.empty((n, 2), dtype=torch.float64, device=device)
        t_par = torch.empty(n, dtype=torch.int64, device=device)
        t_dist = torch.empty(n, dtype=torch.float64, device=device)
    for t in (t_xy, t_par, t_dist):
        dist.broadcast(t, src=winner)
    return winner, costs[winner], t_xy.cpu().numpy(), t_par.cpu().numpy(), t_dist.cpu().numpy()
