#!/usr/bin/env python
"""Generates lara_amd/csrc/mc_tables.h: the marching-cubes case table of the TSDF mesh extraction (csrc/tsdf.hip).

The table is DERIVED here, not transcribed: for each of the 256 sign patterns of a cell's 8 corners (bit i set = corner i
is inside, f < 0) the surface inside the cell is found by tracing its boundary over the six faces --
  * a face whose corners change sign along exactly two of its edges carries one segment between those two crossings;
  * a face with four crossings (diagonally opposite corners inside) is ambiguous; it carries two segments, each cutting
    off one of the two INSIDE corners.  The rule reads only the face's own corner signs, so the two cells sharing the face
    draw the same segments: the extracted surface is watertight (the classic Lorensen table is not, in these cases);
  * segments are linked into closed loops over the cell's edge crossings, each loop is oriented so that its normal points
    from the inside corners to the outside ones (towards growing f) and triangulated as a fan whose diagonals stay off
    the cell's faces.
Corner i sits at (i & 1, (i >> 1) & 1, (i >> 2) & 1); edge e joins EDGE_CORNERS[e].  At most 5 triangles per case (asserted).
usage: python tools/gen_mc_tables.py  [--check]   (--check: compare with the committed header instead of writing it)"""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CORNERS = [(i & 1, (i >> 1) & 1, (i >> 2) & 1) for i in range(8)]
EDGE_CORNERS = [(a, b) for a, b in itertools.combinations(range(8), 2) if sum(x != y for x, y in zip(CORNERS[a], CORNERS[b])) == 1]
assert len(EDGE_CORNERS) == 12
EDGE_OF = {frozenset(e): i for i, e in enumerate(EDGE_CORNERS)}
# the six faces as corner 4-cycles (consecutive corners share a cube edge)
FACES = []
for axis in range(3):
    for side in (0, 1):
        cs = [i for i in range(8) if CORNERS[i][axis] == side]
        u, v = [a for a in range(3) if a != axis]
        key = {(CORNERS[i][u], CORNERS[i][v]): i for i in cs}
        FACES.append([key[(0, 0)], key[(1, 0)], key[(1, 1)], key[(0, 1)]])


def case_triangles(case):
    inside = [(case >> i) & 1 for i in range(8)]
    segs = []
    for f in FACES:
        cross = [k for k in range(4) if inside[f[k]] != inside[f[(k + 1) % 4]]]       # face edge k joins f[k], f[k+1]
        edge = lambda k: EDGE_OF[frozenset((f[k], f[(k + 1) % 4]))]
        if len(cross) == 2:
            segs.append((edge(cross[0]), edge(cross[1])))
        elif len(cross) == 4:       # each inside corner f[k] is cut off by the segment between its two face edges k-1 and k
            for k in range(4):
                if inside[f[k]]:
                    segs.append((edge((k - 1) % 4), edge(k)))
    # link the segments into loops: every crossed edge belongs to exactly two faces, i.e. to two segments
    adj = {}
    for a, b in segs:
        adj.setdefault(a, []).append(b)
        adj.setdefault(b, []).append(a)
    assert all(len(v) == 2 for v in adj.values())
    tris, seen = [], set()
    pos = np.array(CORNERS, dtype=np.float64)
    mid = lambda e: 0.5 * (pos[EDGE_CORNERS[e][0]] + pos[EDGE_CORNERS[e][1]])
    for start in sorted(adj):
        if start in seen:
            continue
        loop, prev, cur = [start], None, start
        seen.add(start)
        while True:
            n = adj[cur][0] if adj[cur][0] != prev else adj[cur][1]
            if n == start:
                break
            assert n not in seen
            loop.append(n)
            seen.add(n)
            prev, cur = cur, n
        assert len(loop) >= 3
        # orientation: Newell normal against the direction inside -> outside summed over the loop's crossed edges
        pts = np.array([mid(e) for e in loop])
        normal = sum(np.cross(pts[i], pts[(i + 1) % len(loop)]) for i in range(len(loop)))
        grad = np.zeros(3)
        for e in loop:
            a, b = EDGE_CORNERS[e]
            grad += (pos[b] - pos[a]) if inside[a] else (pos[a] - pos[b])
        if normal @ grad < 0:
            loop = loop[::-1]
        # fan triangulation from an apex whose diagonals all run through the cell's interior: a diagonal between two
        # crossings of the same face would lie IN that face, where the neighbouring cell may draw the same segment
        # (an edge with more than two triangles)
        on_face = lambda e: {i for i, f in enumerate(FACES) if set(EDGE_CORNERS[e]) <= set(f)}
        for r in range(len(loop)):
            rot = loop[r:] + loop[:r]
            if all(not (on_face(rot[0]) & on_face(rot[i])) for i in range(2, len(rot) - 1)):
                break
        else:
            raise AssertionError(f"case {case}: no interior fan for loop {loop}")
        tris += [(rot[0], rot[i], rot[i + 1]) for i in range(1, len(rot) - 1)]
    return tris


def build():
    table = [case_triangles(c) for c in range(256)]
    assert max(len(t) for t in table) <= 5 and not table[0] and not table[255]
    return table


def header(table):
    out = ["// GENERATED by tools/gen_mc_tables.py -- do not edit.  Marching-cubes cases derived by tracing the cell faces;",
           "// corner i = (i & 1, (i >> 1) & 1, (i >> 2) & 1), bit i of the case = corner i inside (f < 0).",
           "#pragma once", "#include <stdint.h>", "",
           "static const uint8_t MC_EDGE_CORNERS[12][2] = {" + ", ".join("{%d, %d}" % e for e in EDGE_CORNERS) + "};",
           "static const uint8_t MC_NTRI[256] = {" + ", ".join(str(len(t)) for t in table) + "};",
           "// up to 5 triangles x 3 edge ids per case, padded with 255",
           "static const uint8_t MC_TRI[256][15] = {"]
    for t in table:
        flat = [e for tri in t for e in tri] + [255] * (15 - 3 * len(t))
        out.append("    {" + ", ".join(str(x) for x in flat) + "},")
    out.append("};")
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    text = header(build())
    path = os.path.join(ROOT, "lara_amd", "csrc", "mc_tables.h")
    if "--check" in sys.argv:
        sys.exit(0 if open(path).read() == text else 1)
    open(path, "w").write(text)
    print("wrote", path)
