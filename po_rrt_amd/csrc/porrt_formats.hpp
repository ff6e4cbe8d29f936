// porrt_formats.hpp -- the on-disk formats either side of the hot path, host code behind the C ABI.
//
//   porrt_read_pgm        the raster the reference's domains open: image::open -> ImageLuma8 (src/map_shelves_io.rs:88-103,
//                         src/map_io.rs:90-105).  `image` 0.23's PNM decoder, restated from its published behaviour (the crate
//                         is not under /root/reference; parity unpinned against it): magic P2 / P5 (P1 / P4
//                         bitmaps become 0 / 255 gray), header tokens separated by whitespace with '#' comments to the end of
//                         the line, exactly one whitespace byte between maxval and a binary raster, samples stored as they
//                         are (no rescaling by maxval), row-major from the top.  maxval > 255 decodes to 16-bit gray and
//                         colour files to RGB, which the reference rejects ("Wrong image format!") -- so does this reader.
//   porrt_graph_*_json    PTOGraph save / load (src/pto_graph.rs:22-118): serde_json's pretty form of
//                         {"nodes": [{"state": [..], "validity_id": n, "parents": [{"id", "validity_id"}..], "children": [..]}..],
//                          "validities": [[bool..]..]}, floats in serde_json's (ryu) shortest round-trip notation.
#pragma once
#include <cctype>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/porrt_hip.h"

namespace porrt_fmt {

inline bool read_file(const char *path, std::vector<uint8_t> &out) {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.insert(out.end(), buf, buf + n);
    fclose(f);
    return true;
}

struct PnmCursor {
    const uint8_t *p;
    size_t n, i = 0;
    // next header token: skips whitespace and '#' comments (to the end of the line)
    bool token(std::string &t) {
        t.clear();
        for (;;) {
            while (i < n && isspace(p[i])) ++i;
            if (i < n && p[i] == '#') { while (i < n && p[i] != '\n' && p[i] != '\r') ++i; continue; }
            break;
        }
        while (i < n && !isspace(p[i]) && p[i] != '#') t.push_back((char)p[i++]);
        return !t.empty();
    }
    bool number(uint64_t &v) {
        std::string t;
        if (!token(t)) return false;
        v = 0;
        for (char c : t) { if (c < '0' || c > '9') return false; v = v * 10 + (uint64_t)(c - '0'); if (v > (1ull << 40)) return false; }
        return true;
    }
};

// 0 ok; PORRT_ERR_INVALID = not a gray image the reference accepts / malformed
// (header_only: W and H alone, nothing allocated.  The raster is only allocated once the bytes that must fill it are known to be there:
// a binary raster's size, or for the ASCII kinds one byte per sample at least -- a 20-byte file cannot ask for 4 GiB.)
inline int decode_pgm(const uint8_t *bytes, size_t n, std::vector<uint8_t> &out, uint32_t &W, uint32_t &H, bool header_only = false) {
    PnmCursor c{bytes, n};
    std::string magic;
    if (!c.token(magic) || magic.size() != 2 || magic[0] != 'P') return PORRT_ERR_INVALID;
    const char kind = magic[1];
    if (kind != '1' && kind != '2' && kind != '4' && kind != '5') return PORRT_ERR_INVALID;        // P3 / P6 / P7: not ImageLuma8
    uint64_t w, h, maxval = 1;
    if (!c.number(w) || !c.number(h) || w == 0 || h == 0 || w > 0xFFFFFFFFull || h > 0xFFFFFFFFull || w * h > (1ull << 32)) return PORRT_ERR_INVALID;
    if (kind == '2' || kind == '5') { if (!c.number(maxval) || maxval == 0 || maxval > 65535) return PORRT_ERR_INVALID; }
    if (maxval > 255) return PORRT_ERR_INVALID;                                                       // ImageLuma16: "Wrong image format!"
    W = (uint32_t)w; H = (uint32_t)h;
    const size_t np = (size_t)w * h;
    if (header_only) return PORRT_OK;
    if (kind == '5') {                       // one whitespace byte, then the samples
        if (c.i >= n || !isspace(bytes[c.i])) return PORRT_ERR_INVALID;
        ++c.i;
        if (n - c.i < np) return PORRT_ERR_INVALID;
        out.assign(bytes + c.i, bytes + c.i + np);
        return PORRT_OK;
    }
    if (kind == '4') {                       // packed bits, rows padded to bytes, 1 = black
        if (c.i >= n || !isspace(bytes[c.i])) return PORRT_ERR_INVALID;
        ++c.i;
        const size_t stride = (w + 7) / 8;
        if (n - c.i < stride * h) return PORRT_ERR_INVALID;
        out.resize(np);
        for (size_t y = 0; y < h; ++y)
            for (size_t x = 0; x < w; ++x) out[y * w + x] = ((bytes[c.i + y * stride + x / 8] >> (7 - x % 8)) & 1) ? 0 : 255;
        return PORRT_OK;
    }
    if (n - c.i < np) return PORRT_ERR_INVALID;      // ASCII kinds: a sample is one byte at least
    out.resize(np);
    if (kind == '1') {                       // ASCII bits, whitespace optional between them
        size_t k = 0;
        while (k < np && c.i < n) {
            const uint8_t ch = bytes[c.i++];
            if (ch == '0' || ch == '1') out[k++] = ch == '1' ? 0 : 255;
            else if (!isspace(ch)) return PORRT_ERR_INVALID;
        }
        return k == np ? PORRT_OK : PORRT_ERR_INVALID;
    }
    for (size_t k = 0; k < np; ++k) {        // P2: decimal samples separated by whitespace
        while (c.i < n && isspace(bytes[c.i])) ++c.i;
        uint64_t v = 0;
        size_t digits = 0;
        while (c.i < n && bytes[c.i] >= '0' && bytes[c.i] <= '9') { v = v * 10 + (uint64_t)(bytes[c.i++] - '0'); if (++digits > 5) return PORRT_ERR_INVALID; }
        if (!digits || v > 255 || (c.i < n && !isspace(bytes[c.i]))) return PORRT_ERR_INVALID;
        out[k] = (uint8_t)v;
    }
    return PORRT_OK;
}

// f64 in serde_json's notation (ryu): shortest digits that round-trip; plain decimal for 1e-5 <= |x| < 1e16 with at
// least one digit after the point, d.ddde[-]x outside; serde_json writes null for non-finite values.
inline std::string json_f64(double v) {
    if (!std::isfinite(v)) return "null";
    if (v == 0.0) return std::signbit(v) ? "-0.0" : "0.0";
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);      // shortest round-trip digits
    std::string s(buf, r.ptr);
    std::string out;
    size_t p = 0;
    if (s[0] == '-') { out = "-"; p = 1; }
    const size_t e = s.find('e');
    std::string digits;
    for (size_t i = p; i < e; ++i) if (s[i] != '.') digits.push_back(s[i]);
    const int exp10 = atoi(s.c_str() + e + 1);
    const int kk = exp10 + 1;                                    // position of the decimal point relative to the digits
    const int nd = (int)digits.size();
    if (nd <= kk && kk <= 16) {                                  // integer value: digits, zeros, ".0"
        out += digits + std::string((size_t)(kk - nd), '0') + ".0";
    } else if (0 < kk && kk <= 16) {                             // point inside the digits
        out += digits.substr(0, (size_t)kk) + "." + digits.substr((size_t)kk);
    } else if (-5 < kk && kk <= 0) {                             // 0.000ddd
        out += "0." + std::string((size_t)(-kk), '0') + digits;
    } else {                                                     // d.ddde-x / de21
        out += digits.substr(0, 1);
        if (nd > 1) out += "." + digits.substr(1);
        out += "e" + std::to_string(kk - 1);
    }
    return out;
}

struct GraphFile {
    std::vector<double> xy;
    std::vector<uint64_t> node_validity;
    std::vector<uint64_t> coff, poff;              // CSR offsets (n + 1)
    std::vector<uint64_t> cid, cval, pid, pval;
    std::vector<uint8_t> validities;               // [n_validities][n_worlds] as 0 / 1
    uint64_t n_validities = 0, n_worlds = 0;
    std::string err;
};

inline void write_edges(std::string &o, const char *key, const uint64_t *off, const uint64_t *ids, const uint64_t *vals, uint64_t node, bool last) {
    const uint64_t a = off[node], b = off[node + 1];
    o += "      \""; o += key; o += "\": [";
    if (a == b) o += "]";
    else {
        o += "\n";
        for (uint64_t e = a; e < b; ++e) {
            o += "        {\n          \"id\": " + std::to_string(ids[e]) + ",\n          \"validity_id\": " + std::to_string(vals[e]) + "\n        }";
            o += e + 1 < b ? ",\n" : "\n";
        }
        o += "      ]";
    }
    o += last ? "\n" : ",\n";
}

inline std::string graph_to_json(uint64_t n, const double *xy, const uint64_t *nv, const uint64_t *coff, const uint64_t *cid, const uint64_t *cval,
                                 const uint64_t *poff, const uint64_t *pid, const uint64_t *pval, uint64_t n_validities, uint64_t n_worlds,
                                 const uint8_t *validities) {
    std::string o = "{\n  \"nodes\": [";
    if (n == 0) o += "]";
    else {
        o += "\n";
        for (uint64_t i = 0; i < n; ++i) {
            o += "    {\n      \"state\": [\n        " + json_f64(xy[2 * i]) + ",\n        " + json_f64(xy[2 * i + 1]) + "\n      ],\n";
            o += "      \"validity_id\": " + std::to_string(nv[i]) + ",\n";
            write_edges(o, "parents", poff, pid, pval, i, false);
            write_edges(o, "children", coff, cid, cval, i, true);
            o += i + 1 < n ? "    },\n" : "    }\n";
        }
        o += "  ]";
    }
    o += ",\n  \"validities\": [";
    if (n_validities == 0) o += "]";
    else {
        o += "\n";
        for (uint64_t v = 0; v < n_validities; ++v) {
            o += "    [";
            if (n_worlds == 0) o += "]";
            else {
                o += "\n";
                for (uint64_t w = 0; w < n_worlds; ++w) { o += validities[v * n_worlds + w] ? "      true" : "      false"; o += w + 1 < n_worlds ? ",\n" : "\n"; }
                o += "    ]";
            }
            o += v + 1 < n_validities ? ",\n" : "\n";
        }
        o += "  ]";
    }
    o += "\n}";
    return o;
}

// ---- a small JSON reader, enough for what serde_json::from_reader accepts for SerializablePTOGraph (any whitespace, any
// key order, unknown keys ignored)
struct JsonIn {
    const char *p, *end;
    std::string err;
    void ws() { while (p < end && isspace((unsigned char)*p)) ++p; }
    bool eat(char c) { ws(); if (p < end && *p == c) { ++p; return true; } return false; }
    bool expect(char c) { if (eat(c)) return true; if (err.empty()) err = std::string("expected '") + c + "'"; return false; }
    bool str(std::string &s) {
        ws();
        if (p >= end || *p != '"') { if (err.empty()) err = "expected a string"; return false; }
        ++p; s.clear();
        while (p < end && *p != '"') { if (*p == '\\' && p + 1 < end) ++p; s.push_back(*p++); }
        if (p >= end) { err = "unterminated string"; return false; }
        ++p;
        return true;
    }
    bool num(double &v) {
        ws();
        const char *q = p;
        while (q < end && (isdigit((unsigned char)*q) || *q == '-' || *q == '+' || *q == '.' || *q == 'e' || *q == 'E')) ++q;
        if (q == p) { if (err.empty()) err = "expected a number"; return false; }
        auto r = std::from_chars(p, q, v);
        if (r.ec != std::errc()) { err = "bad number"; return false; }
        p = q;
        return true;
    }
    bool uint(uint64_t &v) { double d; if (!num(d)) return false; if (d < 0 || d != std::floor(d)) { err = "expected an unsigned integer"; return false; } v = (uint64_t)d; return true; }
    bool boolean(bool &b) {
        ws();
        if (end - p >= 4 && !strncmp(p, "true", 4)) { b = true; p += 4; return true; }
        if (end - p >= 5 && !strncmp(p, "false", 5)) { b = false; p += 5; return true; }
        if (err.empty()) err = "expected true / false";
        return false;
    }
    bool skip_value(int depth = 0) {     // unknown key (nesting is bounded: the input is a file from anywhere)
        ws();
        if (p >= end) return false;
        if (depth > 128) { err = "JSON nested too deeply"; return false; }
        if (*p == '"') { std::string s; return str(s); }
        if (*p == '{' || *p == '[') {
            const char open = *p, close = open == '{' ? '}' : ']';
            ++p;
            if (eat(close)) return true;
            for (;;) {
                if (open == '{') { std::string k; if (!str(k) || !expect(':')) return false; }
                if (!skip_value(depth + 1)) return false;
                if (eat(',')) continue;
                return expect(close);
            }
        }
        bool b; double d;
        if (*p == 't' || *p == 'f') return boolean(b);
        if (end - p >= 4 && !strncmp(p, "null", 4)) { p += 4; return true; }
        return num(d);
    }
};

inline bool parse_edges(JsonIn &in, std::vector<uint64_t> &ids, std::vector<uint64_t> &vals) {
    if (!in.expect('[')) return false;
    if (in.eat(']')) return true;
    for (;;) {
        if (!in.expect('{')) return false;
        uint64_t id = 0, v = 0;
        bool hid = false, hv = false;
        if (!in.eat('}')) for (;;) {
            std::string k;
            if (!in.str(k) || !in.expect(':')) return false;
            if (k == "id") { if (!in.uint(id)) return false; hid = true; }
            else if (k == "validity_id") { if (!in.uint(v)) return false; hv = true; }
            else if (!in.skip_value()) return false;
            if (in.eat(',')) continue;
            if (!in.expect('}')) return false;
            break;
        }
        if (!hid || !hv) { in.err = "edge without id / validity_id"; return false; }
        ids.push_back(id); vals.push_back(v);
        if (in.eat(',')) continue;
        return in.expect(']');
    }
}

inline bool parse_graph(const std::string &text, GraphFile &g) {
    JsonIn in{text.data(), text.data() + text.size(), ""};
    auto fail = [&]() { g.err = in.err.empty() ? "malformed JSON" : in.err; return false; };
    if (!in.expect('{')) return fail();
    bool have_nodes = false, have_val = false;
    g.coff.assign(1, 0); g.poff.assign(1, 0);
    if (!in.eat('}')) for (;;) {
        std::string key;
        if (!in.str(key) || !in.expect(':')) return fail();
        if (key == "nodes") {
            have_nodes = true;
            if (!in.expect('[')) return fail();
            if (!in.eat(']')) for (;;) {
                if (!in.expect('{')) return fail();
                bool hs = false, hv = false, hp = false, hc = false;
                if (!in.eat('}')) for (;;) {
                    std::string k;
                    if (!in.str(k) || !in.expect(':')) return fail();
                    if (k == "state") {
                        if (!in.expect('[')) return fail();
                        double x, y;
                        if (!in.num(x) || !in.expect(',') || !in.num(y) || !in.expect(']')) { if (in.err.empty()) in.err = "state must hold two numbers"; return fail(); }   // PTOGraph<2>
                        g.xy.push_back(x); g.xy.push_back(y); hs = true;
                    } else if (k == "validity_id") { uint64_t v; if (!in.uint(v)) return fail(); g.node_validity.push_back(v); hv = true; }
                    else if (k == "parents") { if (!parse_edges(in, g.pid, g.pval)) return fail(); hp = true; }
                    else if (k == "children") { if (!parse_edges(in, g.cid, g.cval)) return fail(); hc = true; }
                    else if (!in.skip_value()) return fail();
                    if (in.eat(',')) continue;
                    if (!in.expect('}')) return fail();
                    break;
                }
                if (!hs || !hv || !hp || !hc) { in.err = "node without state / validity_id / parents / children"; return fail(); }
                g.coff.push_back(g.cid.size()); g.poff.push_back(g.pid.size());
                if (in.eat(',')) continue;
                if (!in.expect(']')) return fail();
                break;
            }
        } else if (key == "validities") {
            have_val = true;
            if (!in.expect('[')) return fail();
            if (!in.eat(']')) for (;;) {
                if (!in.expect('[')) return fail();
                uint64_t nw = 0;
                if (!in.eat(']')) for (;;) {
                    bool b;
                    if (!in.boolean(b)) return fail();
                    g.validities.push_back(b ? 1 : 0); ++nw;
                    if (in.eat(',')) continue;
                    if (!in.expect(']')) return fail();
                    break;
                }
                if (g.n_validities && nw != g.n_worlds) { in.err = "validities of different lengths"; return fail(); }
                g.n_worlds = nw; ++g.n_validities;
                if (in.eat(',')) continue;
                if (!in.expect(']')) return fail();
                break;
            }
        } else if (!in.skip_value()) return fail();
        if (in.eat(',')) continue;
        if (!in.expect('}')) return fail();
        break;
    }
    if (!have_nodes || !have_val) { g.err = "missing field `nodes` / `validities`"; return false; }
    return true;
}

} // namespace porrt_fmt

struct porrt_graph_file { porrt_fmt::GraphFile g; };

extern "C" {

int porrt_read_pgm_mem(const uint8_t *bytes, size_t n, uint8_t *out, uint32_t *W, uint32_t *H) {
    if (!bytes || !W || !H) return PORRT_ERR_INVALID;
    return abi_guard([&]() -> int {
        std::vector<uint8_t> px;
        uint32_t w = 0, h = 0;
        const int r = porrt_fmt::decode_pgm(bytes, n, px, w, h, /*header_only=*/out == nullptr);      // the size query reads the header alone
        if (r) return r;
        *W = w; *H = h;
        if (out) memcpy(out, px.data(), px.size());
        return PORRT_OK;
    });
}

int porrt_read_pgm(const char *path, uint8_t *out, uint32_t *W, uint32_t *H) {
    if (!path) return PORRT_ERR_INVALID;
    return abi_guard([&]() -> int {
        std::vector<uint8_t> bytes;
        if (!porrt_fmt::read_file(path, bytes)) return PORRT_ERR_IO;          // "Impossible to open image"
        return porrt_read_pgm_mem(bytes.data(), bytes.size(), out, W, H);
    });
}

static int graph_write_json_impl(const char *path, uint64_t n_nodes, const double *xy, const uint64_t *node_validity, const uint64_t *child_off,
                           const uint64_t *child_id, const uint64_t *child_validity, const uint64_t *parent_off, const uint64_t *parent_id,
                           const uint64_t *parent_validity, uint64_t n_validities, uint64_t n_worlds, const uint8_t *validities) {
    if (!path || (n_nodes && (!xy || !node_validity || !child_off || !parent_off))) return PORRT_ERR_INVALID;
    const std::string s = porrt_fmt::graph_to_json(n_nodes, xy, node_validity, child_off, child_id, child_validity, parent_off, parent_id, parent_validity,
                                                   n_validities, n_worlds, validities);
    FILE *f = fopen(path, "wb");
    if (!f) return PORRT_ERR_IO;
    const bool ok = fwrite(s.data(), 1, s.size(), f) == s.size();
    return (fclose(f) == 0 && ok) ? PORRT_OK : PORRT_ERR_IO;
}

// The PTO graph (or PRM roadmap) of the context's last grow as the reference's JSON.  Adjacency lists as the reference
// holds them (pto.rs:111-120: a node's neighbours at its creation in kd pre-order, then the later nodes that chose it,
// ascending; children and parents are the same list).
int porrt_graph_write_json(const char *path, uint64_t n_nodes, const double *xy, const uint64_t *node_validity, const uint64_t *child_off,
                           const uint64_t *child_id, const uint64_t *child_validity, const uint64_t *parent_off, const uint64_t *parent_id,
                           const uint64_t *parent_validity, uint64_t n_validities, uint64_t n_worlds, const uint8_t *validities) {
    return abi_guard([&]() { return graph_write_json_impl(path, n_nodes, xy, node_validity, child_off, child_id, child_validity, parent_off, parent_id, parent_validity,
                                                          n_validities, n_worlds, validities); });
}

static int graph_save_json_impl(const porrt_ctx *ctx, const char *path);
int porrt_graph_save_json(const porrt_ctx *ctx, const char *path) { return abi_guard([&]() { return graph_save_json_impl(ctx, path); }); }

static int graph_save_json_impl(const porrt_ctx *ctx, const char *path) {
    if (!ctx || !path) return PORRT_ERR_INVALID;
    const uint64_t n = porrt_num_nodes(ctx), ne = porrt_num_edges(ctx);
    if (!n) return PORRT_ERR_INVALID;
    std::vector<double> xy(2 * n);
    int r = porrt_get_tree(ctx, xy.data(), nullptr, nullptr);
    if (r) return r;
    std::vector<uint32_t> nv32(n, 0), ef(ne), et(ne), ev(ne);
    r = porrt_get_node_validity(ctx, nv32.data());
    if (r) return r;
    if (ne) { r = porrt_get_edges(ctx, ef.data(), et.data(), ev.data()); if (r) return r; }
    std::vector<uint64_t> nv(nv32.begin(), nv32.end()), off(n + 1, 0), ids(2 * ne), vals(2 * ne);
    for (uint64_t e = 0; e < ne; ++e) { ++off[et[e] + 1]; ++off[ef[e] + 1]; }
    for (uint64_t i = 0; i < n; ++i) off[i + 1] += off[i];
    std::vector<uint64_t> cur(off.begin(), off.end() - 1);
    // a node's own neighbours first (its forward edges neighbour -> node, in the reference's order) ...
    for (uint64_t e = 0; e < ne; ++e) { const uint64_t at = cur[et[e]]++; ids[at] = ef[e]; vals[at] = ev[e]; }
    // ... then the later nodes that connected to it
    for (uint64_t e = 0; e < ne; ++e) { const uint64_t at = cur[ef[e]]++; ids[at] = et[e]; vals[at] = ev[e]; }
    const int nw = porrt_n_worlds(ctx);
    std::vector<uint64_t> masks(65, 0);
    const int nval = porrt_get_validities(ctx, masks.data());
    if (nval < 0) return nval;
    std::vector<uint8_t> vb((size_t)nval * (size_t)nw);
    for (int v = 0; v < nval; ++v) for (int w = 0; w < nw; ++w) vb[(size_t)v * nw + w] = (masks[v] >> w) & 1ull;
    return porrt_graph_write_json(path, n, xy.data(), nv.data(), off.data(), ids.data(), vals.data(), off.data(), ids.data(), vals.data(), (uint64_t)nval,
                                  (uint64_t)nw, vb.data());
}

porrt_graph_file *porrt_graph_load_json(const char *path, char *err, size_t err_cap) {
    auto fail = [&](const std::string &m) -> porrt_graph_file * { if (err && err_cap) { snprintf(err, err_cap, "%s", m.c_str()); } return nullptr; };
    if (!path) return fail("null path");
    porrt_graph_file *g = nullptr;
    try {
        std::vector<uint8_t> bytes;
        if (!porrt_fmt::read_file(path, bytes)) return fail("impossible to open file");
        g = new porrt_graph_file();
        if (!porrt_fmt::parse_graph(std::string(bytes.begin(), bytes.end()), g->g)) { const std::string m = g->g.err; delete g; return fail(m); }
        // what the reference's later code would index out of bounds on
        const uint64_t n = g->g.node_validity.size();
        for (uint64_t v : g->g.cid) if (v >= n) { delete g; return fail("child id out of range"); }
        for (uint64_t v : g->g.pid) if (v >= n) { delete g; return fail("parent id out of range"); }
        return g;
    } catch (...) {
        delete g;
        if (err && err_cap) snprintf(err, err_cap, "out of memory");
        return nullptr;
    }
}
void porrt_graph_file_free(porrt_graph_file *g) { delete g; }
uint64_t porrt_graph_file_num_nodes(const porrt_graph_file *g) { return g ? g->g.node_validity.size() : 0; }
uint64_t porrt_graph_file_num_children(const porrt_graph_file *g) { return g ? g->g.cid.size() : 0; }
uint64_t porrt_graph_file_num_parents(const porrt_graph_file *g) { return g ? g->g.pid.size() : 0; }
uint64_t porrt_graph_file_num_validities(const porrt_graph_file *g) { return g ? g->g.n_validities : 0; }
uint64_t porrt_graph_file_num_worlds(const porrt_graph_file *g) { return g ? g->g.n_worlds : 0; }
int porrt_graph_file_get(const porrt_graph_file *g, double *xy, uint64_t *node_validity, uint64_t *child_off, uint64_t *child_id, uint64_t *child_validity,
                         uint64_t *parent_off, uint64_t *parent_id, uint64_t *parent_validity, uint8_t *validities) {
    if (!g) return PORRT_ERR_INVALID;
    const porrt_fmt::GraphFile &f = g->g;
    auto cp = [](auto *dst, const auto &src) { if (dst && !src.empty()) memcpy(dst, src.data(), src.size() * sizeof(src[0])); };
    cp(xy, f.xy); cp(node_validity, f.node_validity); cp(child_off, f.coff); cp(child_id, f.cid); cp(child_validity, f.cval);
    cp(parent_off, f.poff); cp(parent_id, f.pid); cp(parent_validity, f.pval); cp(validities, f.validities);
    return PORRT_OK;
}

} // extern "C"
