"""Wavefront OBJ -> (vertices f32 [n,3], indices u32 [m,3]) for `engine.Scene(..., meshes=[...])` (extension: the reference
has no mesh input).  Only `v` and `f` records are read; polygons are fanned into triangles; negative (relative) indices and
`v/vt/vn` index triples are accepted."""
import numpy as np


def load_obj(path):
    verts, tris = [], []
    with open(path, "r", encoding="utf-8", errors="replace") as f:
        for ln, line in enumerate(f, 1):
            rec = line.split("#", 1)[0].split()
            if not rec:
                continue
            if rec[0] == "v":
                if len(rec) < 4:
                    raise ValueError(f"{path}:{ln}: vertex needs three coordinates")
                verts.append([float(rec[1]), float(rec[2]), float(rec[3])])
            elif rec[0] == "f":
                idx = []
                for tok in rec[1:]:
                    i = int(tok.split("/", 1)[0])
                    i = i - 1 if i > 0 else len(verts) + i
                    if not 0 <= i < len(verts):
                        raise ValueError(f"{path}:{ln}: vertex index {tok} out of range")
                    idx.append(i)
                if len(idx) < 3:
                    raise ValueError(f"{path}:{ln}: face needs at least three vertices")
                tris.extend([idx[0], idx[k], idx[k + 1]] for k in range(1, len(idx) - 1))
    return np.array(verts, np.float32).reshape(-1, 3), np.array(tris, np.uint32).reshape(-1, 3)


def save_obj(path, vertices, indices):
    with open(path, "w", encoding="utf-8") as f:
        for v in np.asarray(vertices, np.float32):
            f.write(f"v {float(v[0])!r} {float(v[1])!r} {float(v[2])!r}\n")
        for t in np.asarray(indices, np.uint32):
            f.write(f"f {t[0] + 1} {t[1] + 1} {t[2] + 1}\n")
