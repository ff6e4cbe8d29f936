"""po_rrt_amd -- MI355X-native batched belief-space RRT expansion engine.

Drop-in for the grow/extend hot path of cambyse/po-rrt (RRT::grow_tree, PTO::grow_graph)
behind a C ABI (include/porrt_hip.h, libporrt_hip.so: hand-written HIP for gfx950).
`engine.Engine` is the thin ctypes binding; `planner` mirrors the reference's RRT / PTO
operator interface on top of it.
"""
from .engine import (DOMAIN_DOOR, DOMAIN_SHELF, INCOMPLETE, MODE_PTO, MODE_RRT, OK, Engine, PorrtError,  # noqa: F401
                     conditional_dijkstra, load_library)
