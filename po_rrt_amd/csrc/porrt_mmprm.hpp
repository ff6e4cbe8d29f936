// porrt_mmprm.hpp -- multi-modal PRM growth (second half): MapShelfDomainTampPRM::grow_mm_prm,
// src/map_shelves_tamp_prm.rs:328-393, with ModeTree::{add_mode, add_transition, get_transitions} :135-283,
// sample_observation_of_zone :482-493 and PRM::{grow_graph, add_sample} src/prm.rs:38-109 underneath.
//
// The reference interleaves three things in one sequential loop: which mode (belief) gets the next 190 samples and which
// zones get observation samples (discrete sampler), the mode tree's bookkeeping (new modes and transitions on demand), and
// the growth of one PRM* roadmap per mode with a kd-tree.  Only the last is heavy, and a roadmap is a function of the
// ORDERED list of its nodes alone (edge j -> i, j < i, iff norm2 <= heuristic_radius(i + 1) and the segment is free:
// porrt_prm.hpp).  So the loop runs on the host WITHOUT graphs -- it only decides which point goes to which mode, in which
// order -- and every mode's roadmap is then built at once on the GPU from its point list (k_prm_bin / k_prm_connect, the
// kernels of porrt_grow_prm), edges restored to the reference's adjacency order.  Quirks kept: every mode's sampler is a
// clone of the planner's never-advanced sampler (:212,249), both transitions of a zone are created with observation = true
// (:227,268), the "not there" reaching probability is taken before normalisation (:241).
#pragma once

struct MmMode {
    std::vector<double> belief;
    double reaching_probability = 0;
    std::vector<int> remaining;
    std::vector<int64_t> there, not_there;       // zone -> transition (the two hash maps)
    Pcg64 sampler;                               // PRM::continuous_sampler: a clone per mode
    std::vector<double> xy;                      // the mode's nodes in add_sample order
    std::vector<uint64_t> finals;
    std::vector<uint32_t> efrom, eto;            // forward edges in the reference's order (neighbour -> new node)
};
struct MmTransition {
    uint32_t zone = 0, from = 0, to = 0;
    int observation = 1;
    std::vector<uint64_t> pairs;                 // observation_transitions: [node in from-mode, node in to-mode]
};
struct MmState {
    bool valid = false;
    uint32_t nw = 0;
    uint64_t n_beliefs = 0;
    std::vector<MmMode> modes;
    std::vector<MmTransition> tr;
    double host_s = 0, roadmap_s = 0, device_s = 0;
    void clear() { valid = false; modes.clear(); tr.clear(); }
};

namespace mmprm {
inline bool is_final(const std::vector<double> &b) {             // :19-21
    double m = b[0];
    for (double v : b) if (v >= m) m = v;
    return m >= 0.999;
}
inline void normalize(std::vector<double> &b) {                  // :23-26
    double sum = 0.0;
    for (double v : b) sum = sum + v;
    for (double &v : b) v = v / sum;
}
inline double transition_probability(const std::vector<double> &parent, const std::vector<double> &child) {     // common.rs:187-190
    double s = 0.0;
    for (size_t i = 0; i < parent.size(); ++i) s = s + (child[i] > 0.0 ? parent[i] : 0.0);
    return s;
}
} // namespace mmprm
