"""Minimal PLY / OBJ readers (stand-ins for the absent `plyfile` / `objloader` modules the
reference calls at `src/load.py:80-82,90-94`) and matching writers used by the tests.

PLY: ascii or binary_little_endian; `vertex` element with float32 x,y,z first (extra vertex
properties are skipped), `face` element with one list property of 3 indices.
OBJ: `v x y z` and triangular `f a[/..] b[/..] c[/..]` lines, 1-based (the reference takes
`obj.face[:, 0] - 1`).
"""
import numpy as np

_PLY_TYPES = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4",
              "uint": "u4", "float": "f4", "double": "f8", "int8": "i1", "uint8": "u1",
              "int16": "i2", "uint16": "u2", "int32": "i4", "uint32": "u4",
              "float32": "f4", "float64": "f8"}


def read_ply(path):
    with open(path, "rb") as fh:
        if fh.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, elements = None, []
        while True:
            line = fh.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append((tok[1], int(tok[2]), []))
            elif tok[0] == "property":
                elements[-1][2].append(tok[1:])
            elif tok[0] == "end_header":
                break
        if fmt not in ("ascii", "binary_little_endian"):
            raise NotImplementedError(f"{path}: PLY format {fmt}")
        verts = faces = None
        for name, count, props in elements:
            is_list = any(p[0] == "list" for p in props)
            if fmt == "ascii":
                rows = [fh.readline().split() for _ in range(count)]
                if name == "vertex":
                    verts = np.array([r[:3] for r in rows], dtype=np.float32)
                elif name == "face":
                    faces = np.array([r[1:4] for r in rows], dtype=np.int32)
            elif not is_list:
                dt = np.dtype([(p[1], "<" + _PLY_TYPES[p[0]]) for p in props])
                data = np.frombuffer(fh.read(dt.itemsize * count), dtype=dt)
                if name == "vertex":
                    verts = np.stack([data["x"], data["y"], data["z"]], axis=1).astype(np.float32)
            else:
                if len(props) != 1:
                    raise NotImplementedError("PLY list element with extra properties")
                cnt_t, idx_t = ("<" + _PLY_TYPES[t] for t in props[0][1:3])
                dt = np.dtype([("n", cnt_t), ("v", idx_t, (3,))])
                data = np.frombuffer(fh.read(dt.itemsize * count), dtype=dt)
                if name == "face":
                    if count and not np.all(data["n"] == 3):
                        raise NotImplementedError("PLY with non-triangular faces")
                    faces = data["v"].astype(np.int32)
        if verts is None or faces is None:
            raise ValueError(f"{path}: PLY needs vertex and face elements")
        return verts, faces


def write_ply(path, vertices, faces, binary=True):
    v = np.asarray(vertices, dtype="<f4")
    f = np.asarray(faces, dtype="<i4")
    head = ["ply", f"format {'binary_little_endian' if binary else 'ascii'} 1.0",
            f"element vertex {len(v)}", "property float x", "property float y",
            "property float z", f"element face {len(f)}",
            "property list uchar int vertex_indices", "end_header"]
    with open(path, "wb") as fh:
        fh.write(("\n".join(head) + "\n").encode("ascii"))
        if binary:
            fh.write(v.tobytes())
            rec = np.zeros(len(f), dtype=np.dtype([("n", "u1"), ("v", "<i4", (3,))]))
            rec["n"], rec["v"] = 3, f
            fh.write(rec.tobytes())
        else:
            for p in v:
                fh.write(("%r %r %r\n" % tuple(float(x) for x in p)).encode("ascii"))
            for t in f:
                fh.write(("3 %d %d %d\n" % tuple(int(x) for x in t)).encode("ascii"))


def read_obj(path):
    verts, faces = [], []
    with open(path, "r") as fh:
        for line in fh:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "v":
                verts.append([float(x) for x in tok[1:4]])
            elif tok[0] == "f":
                if len(tok) != 4:
                    raise NotImplementedError("OBJ with non-triangular faces")
                faces.append([int(t.split("/")[0]) - 1 for t in tok[1:4]])
    return np.array(verts, dtype=np.float64), np.array(faces, dtype=np.int32)


def write_obj(path, vertices, faces):
    with open(path, "w") as fh:
        for p in np.asarray(vertices, dtype=np.float64):
            fh.write("v %r %r %r\n" % tuple(float(x) for x in p))
        for t in np.asarray(faces):
            fh.write("f %d %d %d\n" % tuple(int(x) + 1 for x in t))
