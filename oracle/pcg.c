/*
 * pcg.c -- ORACLE (test infrastructure only): restatement of the random stream
 * behind the reference's samplers (src/sample_space.rs:13-36, 45-59):
 *   Pcg64::seed_from_u64(0), Rng::gen_range(l..u) for f64, gen_range(0..n) for usize.
 *
 * The arithmetic lives in third-party crates that are NOT under /root/reference
 * and are only range-pinned by Cargo.toml:9-10 (rand ^0.8.0, rand_pcg ^0.3.0,
 * rand_core 0.6).  It is restated here from the crates' published algorithms:
 *   - rand_pcg 0.3 Lcg128Xsl64 (PCG XSL-RR 128/64): multiplier
 *     0x2360ED051FC65DA44385DF649FCCF645, from_seed reads 4 LE u64 (state lo/hi,
 *     inc lo/hi), inc |= 1, state += inc, step(); next_u64 = step() then
 *     xsl-rr of the new state.
 *   - rand_core 0.6 SeedableRng::seed_from_u64: PCG32 (mul 6364136223846793005,
 *     inc 11634580027462260723) expands the u64 into the 32 seed bytes.
 *   - rand 0.8 UniformFloat<f64>::sample_single: v = (next_u64 >> 12) as a
 *     mantissa in [1,2) minus 1; res = v*scale + low (two roundings); a fresh
 *     value is drawn while res >= high.
 *   - rand 0.8 UniformInt<usize>::sample_single: widening-multiply rejection
 *     with zone = (range << lz(range)) - 1.
 * PARITY UNPINNED against the Rust crates: the reference's own tests only check
 * bounds (sample_space.rs:75-113).  The PCG core is pinned by the generator's
 * published known-answer vectors (pcg64 state 42 / stream 54 and the rand_pcg
 * from_seed vector), see tests/test_oracle_kat.py.  The hot path treats the
 * sample stream as an explicit INPUT, so planner parity never depends on this.
 */
#include "porrt_oracle.h"
#include <string.h>

typedef unsigned __int128 u128;

static const u128 PCG_MULT = (((u128)0x2360ED051FC65DA4ULL) << 64) | (u128)0x4385DF649FCCF645ULL;

static inline void pcg_step(orc_pcg64 *r) { r->state = r->state * PCG_MULT + r->inc; }

static void pcg_from_state_incr(orc_pcg64 *r, u128 state, u128 incr) {
    r->state = state;
    r->inc = incr;
    r->state = r->state + r->inc; /* move away from the initial value */
    pcg_step(r);
}

void orc_pcg64_new(orc_pcg64 *r, u128 state, u128 stream) {
    pcg_from_state_incr(r, state, (stream << 1) | 1);
}

void orc_pcg64_new_u64(orc_pcg64 *r, uint64_t state_lo, uint64_t state_hi, uint64_t stream_lo, uint64_t stream_hi) {
    orc_pcg64_new(r, ((u128)state_hi << 64) | state_lo, ((u128)stream_hi << 64) | stream_lo);
}

void orc_pcg64_get_state(const orc_pcg64 *r, uint64_t out4[4]) {
    out4[0] = (uint64_t)r->state;
    out4[1] = (uint64_t)(r->state >> 64);
    out4[2] = (uint64_t)r->inc;
    out4[3] = (uint64_t)(r->inc >> 64);
}

static uint64_t read_le64(const uint8_t *p) {
    uint64_t v = 0;
    for (int i = 7; i >= 0; --i) v = (v << 8) | p[i];
    return v;
}

void orc_pcg64_from_seed(orc_pcg64 *r, const uint8_t seed[32]) {
    uint64_t s[4];
    for (int i = 0; i < 4; ++i) s[i] = read_le64(seed + 8 * i);
    u128 state = (u128)s[0] | ((u128)s[1] << 64);
    u128 incr = (u128)s[2] | ((u128)s[3] << 64);
    pcg_from_state_incr(r, state, incr | 1);
}

void orc_pcg64_seed_from_u64(orc_pcg64 *r, uint64_t state) {
    const uint64_t MUL = 6364136223846793005ULL, INC = 11634580027462260723ULL;
    uint8_t seed[32];
    for (int chunk = 0; chunk < 8; ++chunk) {
        state = state * MUL + INC;
        uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
        uint32_t rot = (uint32_t)(state >> 59);
        uint32_t x = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
        seed[4 * chunk + 0] = (uint8_t)(x);
        seed[4 * chunk + 1] = (uint8_t)(x >> 8);
        seed[4 * chunk + 2] = (uint8_t)(x >> 16);
        seed[4 * chunk + 3] = (uint8_t)(x >> 24);
    }
    orc_pcg64_from_seed(r, seed);
}

uint64_t orc_pcg64_next_u64(orc_pcg64 *r) {
    pcg_step(r);
    u128 s = r->state;
    uint32_t rot = (uint32_t)(s >> 122);
    uint64_t xsl = (uint64_t)(s >> 64) ^ (uint64_t)s;
    return (xsl >> rot) | (xsl << ((64 - rot) & 63));
}

/* LCG jump-ahead (Brown, "Random number generation with arbitrary strides"):
 * used by the product's device sampler; kept here so tests can pin it. */
void orc_pcg64_advance(orc_pcg64 *r, u128 delta) {
    u128 acc_mult = 1, acc_plus = 0, cur_mult = PCG_MULT, cur_plus = r->inc;
    while (delta > 0) {
        if (delta & 1) {
            acc_mult *= cur_mult;
            acc_plus = acc_plus * cur_mult + cur_plus;
        }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta >>= 1;
    }
    r->state = acc_mult * r->state + acc_plus;
}

static double u64_bits_to_f64(uint64_t b) {
    double d;
    memcpy(&d, &b, 8);
    return d;
}

/* rand 0.8 UniformFloat<f64>::sample_single (used by gen_range(l..u),
 * sample_space.rs:33) */
double orc_gen_range_f64(orc_pcg64 *r, double low, double high) {
    double scale = high - low;
    for (;;) {
        uint64_t bits = orc_pcg64_next_u64(r) >> 12;
        double value1_2 = u64_bits_to_f64(bits | 0x3FF0000000000000ULL);
        double value0_1 = value1_2 - 1.0;
        volatile double prod = value0_1 * scale; /* two roundings, never fused */
        double res = prod + low;
        if (res < high) return res;
        /* rand 0.8: a finite scale is kept and a fresh value is drawn (the
         * scale is only lowered when high-low overflowed to infinity, which the
         * reference's finite sampling boxes never do) */
    }
}

/* rand 0.8 UniformInt<usize>::sample_single -> sample_single_inclusive(low, high-1)
 * (used by gen_range(0..n), sample_space.rs:58) */
uint64_t orc_gen_range_usize(orc_pcg64 *r, uint64_t low, uint64_t high) {
    uint64_t range = (high - 1) - low + 1;
    if (range == 0) return orc_pcg64_next_u64(r);
    uint64_t zone = (range << __builtin_clzll(range)) - 1;
    for (;;) {
        uint64_t v = orc_pcg64_next_u64(r);
        u128 m = (u128)v * (u128)range;
        uint64_t hi = (uint64_t)(m >> 64), lo = (uint64_t)m;
        if (lo <= zone) return low + hi;
    }
}
