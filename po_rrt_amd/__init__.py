"""po_rrt_amd -- MI355X-native batched belief-space RRT expansion engine.

Drop-in for the grow/extend hot path of cambyse/po-rrt (RRT::grow_tree, PTO::grow_graph)
behind a C ABI (include/porrt_hip.h, libporrt_hip.so: hand-written HIP for gfx950).
`engine.Engine` is the thin ctypes binding of a context, `engine.Comm` that of the one exchange of a
query-sharded job (RCCL); `sharding` holds the query partition.  The C++ mirror of the reference's
RRT / PTO / PRM interface is include/porrt.hpp.
"""
from .engine import (DOMAIN_DOOR, DOMAIN_SHELF, INCOMPLETE, MODE_PTO, MODE_RRT, OK, Comm, Engine, PorrtError,  # noqa: F401
                     conditional_dijkstra, exchange_decide, load_library)
