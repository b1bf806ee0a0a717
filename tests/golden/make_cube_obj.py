#!/usr/bin/env python3
"""Writes tests/golden/c3_cube.obj: the cube mesh of BASELINE config 3 as a Wavefront OBJ in
the dialect the loader has to cope with (comment lines, an object name, a material reference,
normals, quads given as v//vn index pairs, smoothing-group lines, blank lines, a trailing
comment).  Geometry: side 2, centred at the origin, three coordinates off by 1e-6 -- the corner
table of raytracer.c_amd/host/scenes.c, i.e. the values of the reference's own cube asset."""
import os

CORNERS = [(1, -1, -1), (1, -1, 1), (-1, -1, 1), (-1, -1, -1),
           (1, 1, -0.999999), (0.999999, 1, 1.000001), (-1, 1, 1), (-1, 1, -1)]
NORMALS = [(0, -1, 0), (0, 1, 0), (1, 0, 0), (0, 0, 1), (-1, 0, 0), (0, 0, -1)]
QUADS = [(1, 2, 3, 4), (5, 8, 7, 6), (1, 5, 6, 2), (2, 6, 7, 3), (3, 7, 8, 4), (5, 1, 4, 8)]


def num(x):
    return ("%.6f" % x).rstrip("0").rstrip(".") if x != int(x) else str(int(x))


def main():
    out = ["# config 3 test mesh: a cube as quads with per-face normals (written by make_cube_obj.py)", "",
           "mtllib none.mtl", "o c3_cube"]
    out += ["v " + " ".join(num(c) for c in p) for p in CORNERS]
    out += [""] + ["vn " + " ".join(num(c) for c in n) for n in NORMALS]
    out += ["usemtl plain", "s off"]
    out += ["f " + " ".join(f"{v}//{k + 1}" for v in q) for k, q in enumerate(QUADS)]
    out += ["", "# end"]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c3_cube.obj")
    open(path, "w").write("\n".join(out) + "\n")
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
