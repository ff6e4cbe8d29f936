#!/usr/bin/env python3
"""Generate the synthetic 8-bit maps committed under tests/golden/maps/ (200x200, and since round 3 a 400x400 and a 300x200 one).

Every .pgm/.png under the reference's data/ is a Git-LFS pointer, so
the benchmark rasters are unavailable; these stand-ins keep the reference's
conventions (map_shelves_io.rs:150-156: 255 free, 127..254 low obstacle = shelf,
0..126 high obstacle = wall; zone rasters: 255 = no zone, k = zone k;
map_io.rs:165-174: 0 obstacle, 255 free, anything else = door whose zone id comes
from the zone raster) and the start/goal coordinates used by the reference's
drivers (src/main.rs:486-848, src/pto.rs:466-540).  They are labelled synthetic
everywhere they are used.

The one real raster recoverable offline is the 200x200 door map embedded (base64
PNG) in data/maps_paper/map_4/map.svg; `--paper-map` re-extracts it when
/root/reference is present (it is data, not source).  Its zone raster is an LFS
pointer, so door zones are labelled here by connected components.

Run from the repo root:  python tools/make_maps.py
"""
import argparse
import base64
import io
import os
import re

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "maps")
W = H = 200
LOW, UP = -1.0, 1.0
PPM = W / (UP - LOW)


def to_pixel(x, y):
    """map_shelves_io.rs:165-170 (for in-range points)."""
    i = int((H - 1) - (y - LOW) * PPM)
    j = int((x - LOW) * PPM)
    return max(i, 0), max(j, 0)


def rect(a, x0, y0, x1, y1, v):
    i0, j0 = to_pixel(x0, y1)
    i1, j1 = to_pixel(x1, y0)
    a[max(i0, 0):min(i1, H - 1) + 1, max(j0, 0):min(j1, W - 1) + 1] = v


def write_pgm(path, a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    with open(path, "wb") as f:
        f.write(b"P5\n# synthetic map, tools/make_maps.py\n%d %d\n255\n" % (a.shape[1], a.shape[0]))
        f.write(a.tobytes())


def read_pgm(path):
    with open(path, "rb") as f:
        data = f.read()
    # P5 with optional comments
    toks, pos = [], 2
    assert data[:2] == b"P5"
    while len(toks) < 3:
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            pos = data.index(b"\n", pos) + 1
            continue
        end = pos
        while not data[end:end + 1].isspace():
            end += 1
        toks.append(int(data[pos:end]))
        pos = end
    pos += 1
    w, h, _ = toks
    return np.frombuffer(data[pos:pos + w * h], dtype=np.uint8).reshape(h, w).copy()


def clear_disk(a, x, y, r):
    for i in range(H):
        for j in range(W):
            cx = j / PPM + LOW + 0.5 / PPM
            cy = (H - 1 - i) / PPM + LOW + 0.5 / PPM
            if abs(cx - x) + abs(cy - y) <= r:
                a[i, j] = 255


def map0_like():
    """One room with two interior walls (cfg1 plumbing; start (0,0), goal (0,0.9))."""
    a = np.full((H, W), 255, np.uint8)
    rect(a, -0.6, 0.35, 0.45, 0.40, 0)
    rect(a, -0.45, -0.45, -0.40, 0.2, 0)
    rect(a, 0.3, -0.7, 0.35, -0.1, 0)
    clear_disk(a, 0.0, 0.0, 0.08)
    clear_disk(a, 0.0, 0.9, 0.08)
    return a


def benchmark_like(seed):
    """Perimeter shelves + interior racks and pillars; start (0,-1); goals on the
    perimeter lattice used by main.rs:491-497, 767-774 (+-0.9 / +-0.5 / 0)."""
    rng = np.random.default_rng(1000 + seed)
    a = np.full((H, W), 255, np.uint8)
    shelf = 200
    # perimeter shelves (low obstacles), broken by gaps
    for (x0, x1) in [(-1.0, -0.93), (0.93, 1.0)]:
        for y0 in np.arange(-0.8, 0.8, 0.4):
            rect(a, x0, y0 + 0.06, x1, y0 + 0.34, shelf)
    for x0 in np.arange(-0.8, 0.8, 0.4):
        rect(a, x0 + 0.06, 0.93, x0 + 0.34, 1.0, shelf)
    # interior racks: random axis-aligned low obstacles and a few high pillars
    n_racks = 10 + seed % 3
    for _ in range(n_racks):
        cx, cy = rng.uniform(-0.75, 0.75), rng.uniform(-0.7, 0.75)
        if rng.random() < 0.5:
            w, h = rng.uniform(0.15, 0.35), 0.05
        else:
            w, h = 0.05, rng.uniform(0.15, 0.35)
        rect(a, cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2, shelf if rng.random() < 0.6 else 0)
    for _ in range(6):
        cx, cy = rng.uniform(-0.8, 0.8), rng.uniform(-0.6, 0.8)
        rect(a, cx - 0.03, cy - 0.03, cx + 0.03, cy + 0.03, 0)
    # keep start and goal surroundings free
    clear_disk(a, 0.0, -1.0, 0.12)
    for gx, gy in [(-0.9, -0.5), (-0.9, 0.0), (-0.9, 0.5), (-0.5, 0.9), (0.5, 0.9), (0.9, 0.5), (0.9, 0.0), (0.9, -0.5)]:
        clear_disk(a, gx * 0.92, gy * 0.92, 0.09)
    return a


def benchmark_zone_ids(n):
    """zone k = the shelf cell behind goal k (main.rs:491-497, 767-774 order)."""
    goals = {2: [(-0.9, 0.0), (0.9, 0.0)],
             4: [(-0.9, 0.0), (-0.5, 0.9), (0.5, 0.9), (0.9, 0.0)],
             6: [(-0.9, -0.5), (-0.9, 0.5), (-0.5, 0.9), (0.5, 0.9), (0.9, 0.5), (0.9, -0.5)],
             8: [(-0.9, -0.5), (-0.9, 0.0), (-0.9, 0.5), (-0.5, 0.9), (0.5, 0.9), (0.9, 0.5), (0.9, 0.0), (0.9, -0.5)]}[n]
    z = np.full((H, W), 255, np.uint8)
    for k, (gx, gy) in enumerate(goals):
        sx = np.sign(gx) if abs(gx) > abs(gy) else 0
        sy = np.sign(gy) if abs(gy) >= abs(gx) else 0
        cx, cy = gx + 0.06 * sx, gy + 0.06 * sy
        rect(z, cx - 0.03, cy - 0.03, cx + 0.03, cy + 0.03, k)
    return z, goals


def map1_2_goals_like():
    """Two shelf goals on the right, a dividing wall (pto.rs:466-479)."""
    a = np.full((H, W), 255, np.uint8)
    rect(a, -0.2, -0.15, 0.9, -0.10, 0)     # wall between the two goal corridors
    rect(a, -0.25, -0.6, -0.2, 0.5, 0)
    rect(a, 0.74, -0.55, 0.80, -0.35, 200)  # shelf 0
    rect(a, 0.74, 0.28, 0.80, 0.48, 200)    # shelf 1
    z = np.full((H, W), 255, np.uint8)
    rect(z, 0.75, -0.48, 0.79, -0.42, 0)
    rect(z, 0.75, 0.35, 0.79, 0.41, 1)
    return a, z


def map5_12_goals_like():
    """4x3 lattice of shelves (main.rs:386-408): x in {-0.75,-0.25,0.25,0.75},
    y in {0.75,0.25,-0.25}; goal k sits just below shelf k."""
    a = np.full((H, W), 255, np.uint8)
    z = np.full((H, W), 255, np.uint8)
    goals = []
    k = 0
    for y in (0.75, 0.25, -0.25):
        for x in (-0.75, -0.25, 0.25, 0.75):
            rect(a, x - 0.12, y + 0.07, x + 0.12, y + 0.13, 200)
            rect(z, x - 0.03, y + 0.08, x + 0.03, y + 0.12, k)
            goals.append((x, y))
            k += 1
    rect(a, -0.5, -0.55, 0.5, -0.5, 0)
    return a, z, goals


def door_zone_ids(occ):
    """Label door pixels (not 0 / 255) by 4-connected components in scan order."""
    z = np.full(occ.shape, 255, np.uint8)
    door = (occ != 0) & (occ != 255)
    seen = np.zeros(occ.shape, bool)
    k = 0
    for i in range(occ.shape[0]):
        for j in range(occ.shape[1]):
            if door[i, j] and not seen[i, j]:
                stack = [(i, j)]
                seen[i, j] = True
                while stack:
                    ci, cj = stack.pop()
                    z[ci, cj] = k
                    for di, dj in ((1, 0), (-1, 0), (0, 1), (0, -1)):
                        ni, nj = ci + di, cj + dj
                        if 0 <= ni < occ.shape[0] and 0 <= nj < occ.shape[1] and door[ni, nj] and not seen[ni, nj]:
                            seen[ni, nj] = True
                            stack.append((ni, nj))
                k += 1
    return z, k


def door_map_like():
    """Two rooms split by a wall with two doors (value 128), map_io.rs conventions."""
    a = np.full((H, W), 255, np.uint8)
    rect(a, -1.0, 0.0, 1.0, 0.04, 0)
    rect(a, -0.6, 0.0, -0.45, 0.04, 128)
    rect(a, 0.4, 0.0, 0.55, 0.04, 128)
    rect(a, 0.0, 0.04, 0.04, 0.6, 0)
    return a


class Canvas:
    """A raster of any size over any box with the reference's transform (map_io.rs:176-181, map_shelves_io.rs:165-170:
    ppm = W / (up0 - low0), row i = (H - 1) - (y - low1) ppm, column j = (x - low0) ppm) -- generic in W and H."""

    def __init__(self, W, H, low, up, fill=255):
        self.W, self.H, self.low, self.up = W, H, low, up
        self.ppm = W / (up[0] - low[0])
        self.a = np.full((H, W), fill, np.uint8)

    def px(self, x, y):
        i = int((self.H - 1) - (y - self.low[1]) * self.ppm)
        j = int((x - self.low[0]) * self.ppm)
        return min(max(i, 0), self.H - 1), min(max(j, 0), self.W - 1)

    def rect(self, x0, y0, x1, y1, v):
        i0, j0 = self.px(x0, y1)
        i1, j1 = self.px(x1, y0)
        self.a[i0:i1 + 1, j0:j1 + 1] = v


def big_shelf_map_400():
    """400 x 400 over [-1, 1)^2 (ppm 200: the reference opens 400 x 400 rasters, data/map2_fov.pgm) -- shelf domain, two shelves with
    zones, walls, pillars; a free band wider than 255 pixels nowhere, but clearances up to ~90 pixels."""
    c = Canvas(400, 400, (-1.0, -1.0), (1.0, 1.0))
    z = Canvas(400, 400, (-1.0, -1.0), (1.0, 1.0))
    c.rect(-0.3, -0.2, 0.9, -0.17, 0)
    c.rect(-0.33, -0.6, -0.3, 0.5, 0)
    c.rect(0.2, 0.3, 0.23, 0.8, 0)
    c.rect(-0.8, 0.6, -0.5, 0.63, 200)
    rng = np.random.default_rng(4242)
    for _ in range(14):
        cx, cy = rng.uniform(-0.85, 0.85), rng.uniform(-0.85, 0.85)
        if abs(cx + 0.8) + abs(cy + 0.8) < 0.25:
            continue
        c.rect(cx - 0.02, cy - 0.02, cx + 0.02, cy + 0.02, 0 if rng.random() < 0.5 else 180)
    c.rect(0.74, -0.55, 0.80, -0.35, 200)
    c.rect(0.74, 0.28, 0.80, 0.48, 200)
    z.rect(0.75, -0.48, 0.79, -0.42, 0)
    z.rect(0.75, 0.35, 0.79, 0.41, 1)
    for gx, gy in ((0.68, -0.45), (0.68, 0.38), (-0.8, -0.8)):
        c.rect(gx - 0.04, gy - 0.04, gx + 0.04, gy + 0.04, 255)
    return c.a, z.a


def wide_door_map_300x200():
    """300 x 200 over [-1.5, 1.5) x [-1, 1) (ppm 100, W != H) -- door domain: three rooms side by side, two doors (4 worlds)."""
    c = Canvas(300, 200, (-1.5, -1.0), (1.5, 1.0))
    c.rect(-0.52, -1.0, -0.48, 1.0, 0)
    c.rect(0.48, -1.0, 0.52, 1.0, 0)
    c.rect(-0.52, 0.3, -0.48, 0.5, 128)          # door 0
    c.rect(0.48, -0.5, 0.52, -0.3, 128)          # door 1
    c.rect(-0.52, -0.7, -0.48, -0.6, 255)        # an opening that is always there
    c.rect(0.48, 0.6, 0.52, 0.7, 255)
    c.rect(-0.1, -0.4, -0.06, 0.6, 0)
    c.rect(0.9, 0.0, 1.3, 0.04, 0)
    return c.a


def extract_paper_map():
    from PIL import Image
    svg = "/root/reference/data/maps_paper/map_4/map.svg"
    s = open(svg).read()
    m = re.findall(r'base64,([A-Za-z0-9+/=\s]+)"', s)[0]
    return np.array(Image.open(io.BytesIO(base64.b64decode(m))).convert("L"))


def paper_door_rects(directory="/root/reference/data/maps_paper/map_4"):
    """The reference's own door numbering of map_4: map_door_<k>.svg each hold ONE door's rectangle over the drawing of map.svg
    (viewBox 5421.26 units = the 200 pixels of the raster; the layer is translated by (0, 4368.9), door 3's rectangle sits in a
    nested group translated by (66.39, 638.71)).  Returns, per door k, (x0, x1, y0, y1) in viewBox units -- data read from the
    reference's files, no source text."""
    out = []
    for k in range(4):
        s = open(os.path.join(directory, "map_door_%d.svg" % k)).read()
        s = re.sub(r'base64,[A-Za-z0-9+/=\s]+"', '"', s)
        vb = [float(v) for v in re.search(r'viewBox="([^"]+)"', s).group(1).split()]
        layer = [float(v) for v in re.search(r'id="layer1"\s+transform="translate\(([^)]+)\)"', s).group(1).split(",")]
        # the one green shape of the file: a <rect> (doors 1-3) or a closed <path> of relative cubic segments (door 0)
        g = re.search(r'<g\s+style="opacity:0.5"\s+id="g4168"\s+transform="translate\(([^)]+)\)"[^>/]*>(.*?)</g>', s, re.S)       # (not the self-closing, empty group)
        inner = g.group(2) if g else ""
        off = [float(v) for v in g.group(1).split(",")] if g and "fill:#00ff00" in inner else [0.0, 0.0]
        m = re.search(r'<rect\s+style="[^"]*fill:#00ff00[^"]*"[^>]*?width="([^"]+)"\s+height="([^"]+)"\s+x="([^"]+)"\s+y="([^"]+)"', s, re.S)
        if m:
            w, h, x, y = (float(v) for v in m.groups())
        else:
            m = re.search(r'<path\s+style="[^"]*fill:#00ff00[^"]*"\s+d="m ([^"]+)"', s, re.S)
            tok = m.group(1).replace(",", " ").split()
            x, y = float(tok[0]), float(tok[1])
            nums = [float(t) for t in tok[2:] if t not in ("c", "z")]
            px, py, xs, ys = x, y, [x], [y]
            for q in range(0, len(nums) - 5, 6):          # relative cubic: the third pair is the segment's end point
                px, py = px + nums[q + 4], py + nums[q + 5]
                xs.append(px); ys.append(py)
            x, y, w, h = min(xs), min(ys), max(xs) - min(xs), max(ys) - min(ys)
        x0, y0 = x + off[0] + layer[0], y + off[1] + layer[1]
        out.append((x0 / vb[2], (x0 + w) / vb[2], y0 / vb[3], (y0 + h) / vb[3]))
    return out


def paper_zone_ids(occ, rects):
    """Zone raster of map_4 with the reference's door numbering: door k of map_door_<k>.svg is the component of door pixels whose
    columns its rectangle covers (the drawing and the embedded raster agree in x to a pixel; in y the raster sits about 11 rows
    lower than the drawing, so columns decide -- they are distinct for all four doors).  rects: fractions of the drawing's width."""
    comp, k = door_zone_ids(occ)
    z = np.full(occ.shape, 255, np.uint8)
    used = []
    for door, (fx0, fx1, fy0, fy1) in enumerate(rects):
        best, best_ov = -1, 0.0
        for c in range(k):
            ii, jj = np.where(comp == c)
            ov = max(0.0, min(fx1 * occ.shape[1], jj.max() + 1) - max(fx0 * occ.shape[1], jj.min()))
            ov /= (jj.max() + 1 - jj.min())
            dy = abs((fy0 + fy1) / 2 * occ.shape[0] - (ii.min() + ii.max() + 1) / 2)
            if ov > 0.7 and dy < 20 and ov > best_ov:
                best, best_ov = c, ov
        assert best >= 0 and best not in used, "door %d of the reference's svg matches no component of the raster" % door
        used.append(best)
        z[comp == best] = door
    assert len(used) == k, "the raster holds a door the reference's svgs do not name"
    return z, used


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--paper-map", action="store_true", help="re-extract the raster embedded in the reference's map_4 svg")
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    write_pgm(os.path.join(OUT, "map0_like.pgm"), map0_like())
    write_pgm(os.path.join(OUT, "map_benchmark_like.pgm"), benchmark_like(0))
    for idx, letter in enumerate("abcdefghi"):
        write_pgm(os.path.join(OUT, "map_benchmark_like_%s.pgm" % letter), benchmark_like(idx + 1))
    for n in (2, 4, 6, 8):
        z, _ = benchmark_zone_ids(n)
        write_pgm(os.path.join(OUT, "map_benchmark_like_%d_goals_zone_ids.pgm" % n), z)
    a, z = map1_2_goals_like()
    write_pgm(os.path.join(OUT, "map1_2_goals_like.pgm"), a)
    write_pgm(os.path.join(OUT, "map1_2_goals_like_zone_ids.pgm"), z)
    a, z, _ = map5_12_goals_like()
    write_pgm(os.path.join(OUT, "map5_like.pgm"), a)
    write_pgm(os.path.join(OUT, "map5_like_12_goals_zone_ids.pgm"), z)
    d = door_map_like()
    write_pgm(os.path.join(OUT, "door_map_like.pgm"), d)
    write_pgm(os.path.join(OUT, "door_map_like_zone_ids.pgm"), door_zone_ids(d)[0])
    a, z = big_shelf_map_400()
    write_pgm(os.path.join(OUT, "big_shelf_map_400.pgm"), a)
    write_pgm(os.path.join(OUT, "big_shelf_map_400_zone_ids.pgm"), z)
    d = wide_door_map_300x200()
    write_pgm(os.path.join(OUT, "wide_door_map_300x200.pgm"), d)
    write_pgm(os.path.join(OUT, "wide_door_map_300x200_zone_ids.pgm"), door_zone_ids(d)[0])
    paper = os.path.join(OUT, "paper_map_4.pgm")
    if args.paper_map:
        write_pgm(paper, extract_paper_map())
    doors = os.path.join(OUT, "paper_map_4_doors.json")
    if args.paper_map:                       # (needs /root/reference: the door rectangles are kept as a small data fixture)
        import json
        json.dump({"source": "data/maps_paper/map_4/map_door_{0..3}.svg: the green rectangle of each file as fractions (x0, x1, y0, y1) of the drawing",
                   "rects": paper_door_rects()}, open(doors, "w"), indent=1)
    if os.path.exists(paper) and os.path.exists(doors):
        import json
        occ = read_pgm(paper)
        z, used = paper_zone_ids(occ, json.load(open(doors))["rects"])
        write_pgm(os.path.join(OUT, "paper_map_4_zone_ids.pgm"), z)
        print("paper_map_4: zone k = door k of the reference's map_door_k.svg = connected component (scan order) %s" % used)
    for f in sorted(os.listdir(OUT)):
        if not f.endswith(".pgm"):
            continue
        a = read_pgm(os.path.join(OUT, f))
        print("%-46s free %.1f%%  low %.1f%%  high %.1f%%" % (f, 100 * (a == 255).mean(),
              100 * ((a >= 127) & (a < 255)).mean(), 100 * (a < 127).mean()))


if __name__ == "__main__":
    main()
