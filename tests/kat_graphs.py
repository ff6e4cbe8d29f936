"""The two hand-built belief graphs of the reference's own tests (src/belief_graph.rs:278-500, create_graph_1 /
create_graph_2) as data: node states, belief ids, node types and the add_edge sequence.  Shared by the oracle KATs and
the GPU parity tests."""
A, O = 1, 2          # BeliefNodeType::Action / Observation (belief_graph.rs:13-17)

BELIEFS = [[0.4, 0.6], [1.0, 0.0], [0.0, 1.0]]


def _build(nodes, edges):
    n = len(nodes)
    xy = [list(s) for s, _, _ in nodes]
    belief_id = [b for _, b, _ in nodes]
    types = [t for _, _, t in nodes]
    children = [[] for _ in range(n)]
    parents = [[] for _ in range(n)]
    for a, b in edges:                       # BeliefGraph::add_edge (belief_graph.rs:60-63)
        children[a].append(b)
        parents[b].append(a)
    return dict(xy=xy, belief_id=belief_id, belief_vec=list(belief_id), types=types, children=children, parents=parents, beliefs=BELIEFS)


def graph_1():
    """belief_graph.rs:278-388; finals [3, 10, 16]"""
    nodes = [((0.0, 1.0), 0, A), ((-1.0, 2.0), 0, A), ((1.0, 2.0), 0, A), ((0.0, 4.0), 0, A), ((0.0, 0.0), 0, O),
             ((0.0, 0.0), 1, A), ((0.0, 1.0), 1, A), ((-1.0, 2.0), 1, A), ((1.0, 2.0), 1, A), ((-1.0, 3.0), 1, A), ((0.0, 4.0), 1, A),
             ((0.0, 0.0), 2, A), ((0.0, 1.0), 2, A), ((-1.0, 2.0), 2, A), ((1.0, 2.0), 2, A), ((10.0, 3.0), 2, A), ((0.0, 4.0), 2, A)]
    both = lambda a, b: [(a, b), (b, a)]
    edges = both(0, 1) + both(0, 2) + [(0, 4)]
    edges += [(4, 5)] + both(5, 6) + both(6, 7) + both(6, 8) + both(7, 9) + both(9, 10)
    edges += [(4, 11)] + both(11, 12) + both(12, 13) + both(12, 14) + both(14, 15) + both(15, 16)
    g = _build(nodes, edges)
    g["finals"] = [3, 10, 16]
    return g


def graph_2():
    """belief_graph.rs:390-500; finals [8, 17, 27].  Nodes 18..27 carry belief id 2 with belief_states[1] as their
    vector in the reference (a slip there that the tests do not notice: only the ids enter the clustering, and node 1's
    children 10 and 19 get p = 0.4 each); restated as written."""
    nodes = [((0.0, 0.0), 0, A), ((0.0, 1.0), 0, O), ((1.0, 0.0), 0, A), ((2.0, 0.0), 0, A), ((2.0, 1.0), 0, A), ((2.0, 2.0), 0, A),
             ((2.0, 3.0), 0, A), ((1.0, 3.0), 0, A), ((0.0, 3.0), 0, A),
             ((0.0, 0.0), 1, A), ((0.0, 1.0), 1, A), ((1.0, 0.0), 1, A), ((2.0, 0.0), 1, A), ((2.0, 1.0), 1, A), ((2.0, 2.0), 1, A),
             ((2.0, 3.0), 1, A), ((1.0, 3.0), 1, A), ((0.0, 3.0), 1, A),
             ((0.0, 0.0), 2, A), ((0.0, 1.0), 2, A), ((0.0, 2.0), 2, A), ((1.0, 0.0), 2, A), ((2.0, 0.0), 2, A), ((2.0, 1.0), 2, A),
             ((2.0, 2.0), 2, A), ((2.0, 3.0), 2, A), ((1.0, 3.0), 2, A), ((0.0, 3.0), 2, A)]
    both = lambda a, b: [(a, b), (b, a)]
    edges = [(0, 1)] + both(0, 2) + both(2, 3) + both(3, 4) + both(4, 5) + both(5, 6) + both(6, 7) + both(7, 8)
    edges += [(1, 10)] + both(10, 9) + both(9, 11) + both(11, 12) + both(12, 13) + both(13, 14) + both(14, 15) + both(15, 16) + both(16, 17)
    edges += [(1, 19)] + both(19, 20) + both(20, 27) + both(19, 18) + both(18, 21) + both(21, 22) + both(22, 23) + both(23, 24) + both(24, 25)
    edges += both(26, 25) + both(27, 26)
    g = _build(nodes, edges)
    g["finals"] = [8, 17, 27]
    g["belief_vec"] = [0] * 9 + [1] * 9 + [1] * 10       # nodes 18..27: belief id 2, vector belief_states[1]
    return g
