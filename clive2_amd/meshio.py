"""PLY / OBJ readers (stand-ins for the absent `plyfile` / `objloader` modules the reference calls at
`src/load.py:76-96`) and matching writers used by the tests.

What the reference takes from those libraries is small: `ply["vertex"]` as (n, 3) float32 coordinates and
`ply["face"]["vertex_indices"]` as (m, 3) indices (`load.py:88-94`); `obj.vert` and the vertex index of every face corner,
1-based (`load.py:80-82`).  The readers return exactly that -- (vertices, faces) with faces (m, 3) int32, 0-based -- for
the files those libraries read:

PLY   `format ascii | binary_little_endian | binary_big_endian 1.0`; any elements in any order; the `vertex` element's
      x, y, z wherever they stand among its properties and whatever their scalar type (extra properties -- the Stanford
      bunny's `confidence`, `intensity` -- are skipped; the reference's own `.view(np.float32).reshape(-1, 3)` only works
      for a vertex element of exactly three float32, `load.py:90-92`); the `face` element's list property
      `vertex_indices` / `vertex_index` with any count / index types.  Polygons are fan-triangulated (v0 vi vi+1).
OBJ   `v x y z [w]`; `f` with corners `v`, `v/vt`, `v//vn`, `v/vt/vn`; negative (relative) indices; polygons
      fan-triangulated; every other statement (vt, vn, o, g, s, usemtl, mtllib, comments) is ignored.
Refused with a message that says what and where: truncated data, a face with fewer than three corners, an index outside
the vertex array, a list property in the vertex element, unknown scalar types.
"""
import numpy as np

_PLY_TYPES = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4",
              "uint": "u4", "float": "f4", "double": "f8", "int8": "i1", "uint8": "u1",
              "int16": "i2", "uint16": "u2", "int32": "i4", "uint32": "u4",
              "float32": "f4", "float64": "f8"}
_FACE_LISTS = ("vertex_indices", "vertex_index")


def _ply_type(name, path):
    if name not in _PLY_TYPES:
        raise ValueError(f"{path}: unknown PLY scalar type '{name}'")
    return _PLY_TYPES[name]


def _fan(corners):
    """(m, 3) triangles of polygons given as index lists: v0 vi vi+1."""
    return [(c[0], c[i], c[i + 1]) for c in corners for i in range(1, len(c) - 1)]


def read_ply(path):
    with open(path, "rb") as fh:
        if fh.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, elements = None, []
        while True:
            line = fh.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii", errors="replace").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append((tok[1], int(tok[2]), []))
            elif tok[0] == "property":
                if not elements:
                    raise ValueError(f"{path}: property before any element")
                elements[-1][2].append(tok[1:])
            elif tok[0] == "end_header":
                break
        if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
            raise ValueError(f"{path}: PLY format '{fmt}' (expected ascii, binary_little_endian or binary_big_endian)")
        order = ">" if fmt == "binary_big_endian" else "<"
        verts = faces = None
        for name, count, props in elements:
            lists = [p for p in props if p[0] == "list"]
            if name == "vertex" and lists:
                raise ValueError(f"{path}: list property '{lists[0][-1]}' in the vertex element")
            if fmt == "ascii":
                rows = []
                for k in range(count):
                    line = fh.readline()
                    if not line:
                        raise ValueError(f"{path}: element '{name}' ends after {k} of {count} rows")
                    rows.append(line.split())
                if name == "vertex":
                    verts = _vertex_columns(path, props, rows)
                elif name == "face":
                    faces = _faces_from_rows(path, props, rows)
                continue
            if not lists:
                dt = np.dtype([(p[1], order + _ply_type(p[0], path)) for p in props])
                raw = fh.read(dt.itemsize * count)
                if len(raw) != dt.itemsize * count:
                    raise ValueError(f"{path}: element '{name}' is truncated ({len(raw)} of {dt.itemsize * count} bytes)")
                if name == "vertex":
                    data = np.frombuffer(raw, dtype=dt)
                    for axis in "xyz":
                        if axis not in dt.names:
                            raise ValueError(f"{path}: the vertex element has no '{axis}' property")
                    verts = np.stack([data["x"], data["y"], data["z"]], axis=1).astype(np.float32)
                continue
            corners = _binary_list_element(path, fh, name, count, props, order)
            if name == "face":
                faces = corners
        if verts is None or faces is None:
            raise ValueError(f"{path}: PLY needs a vertex and a face element")
        faces = np.asarray(_fan(faces) if not isinstance(faces, np.ndarray) else faces, dtype=np.int64).reshape(-1, 3)
        _check_indices(path, faces, len(verts))
        return verts, faces.astype(np.int32)


def _vertex_columns(path, props, rows):
    names = [p[1] for p in props]
    try:
        cols = [names.index(a) for a in "xyz"]
    except ValueError:
        raise ValueError(f"{path}: the vertex element needs x, y and z properties (has {names})") from None
    try:
        return np.array([[r[c] for c in cols] for r in rows], dtype=np.float64).astype(np.float32).reshape(-1, 3)
    except (IndexError, ValueError):
        raise ValueError(f"{path}: malformed vertex row") from None


def _faces_from_rows(path, props, rows):
    """ASCII face rows: properties in order, the corner list among them."""
    out = []
    for k, r in enumerate(rows):
        pos, corners = 0, None
        try:
            for p in props:
                if p[0] == "list":
                    n = int(r[pos])
                    items = r[pos + 1:pos + 1 + n]
                    if len(items) != n:
                        raise IndexError
                    pos += 1 + n
                    if p[-1] in _FACE_LISTS:
                        corners = [int(x) for x in items]
                else:
                    pos += 1
        except (IndexError, ValueError):
            raise ValueError(f"{path}: malformed face row {k}") from None
        if corners is None:
            raise ValueError(f"{path}: the face element has no vertex_indices list (properties: {[p[-1] for p in props]})")
        if len(corners) < 3:
            raise ValueError(f"{path}: face {k} has {len(corners)} corners")
        out.append(corners)
    return out


def _binary_list_element(path, fh, name, count, props, order):
    """An element with list properties.  Fast path: one list property and every row three items long (what the reference's
    meshes are); otherwise row by row."""
    if len(props) == 1 and count:
        cnt_t, idx_t = (order + _ply_type(t, path) for t in props[0][1:3])
        dt = np.dtype([("n", cnt_t), ("v", idx_t, (3,))])
        start = fh.tell()
        raw = fh.read(dt.itemsize * count)
        if len(raw) == dt.itemsize * count:
            data = np.frombuffer(raw, dtype=dt)
            if np.all(data["n"] == 3):
                return data["v"].astype(np.int64) if props[0][-1] in _FACE_LISTS else None
        fh.seek(start)
    out = []
    for k in range(count):
        corners = None
        for p in props:
            if p[0] == "list":
                cnt_t, idx_t = (np.dtype(order + _ply_type(t, path)) for t in p[1:3])
                raw = fh.read(cnt_t.itemsize)
                if len(raw) != cnt_t.itemsize:
                    raise ValueError(f"{path}: element '{name}' is truncated in row {k}")
                n = int(np.frombuffer(raw, cnt_t)[0])
                raw = fh.read(idx_t.itemsize * n)
                if len(raw) != idx_t.itemsize * n:
                    raise ValueError(f"{path}: element '{name}' is truncated in row {k}")
                if p[-1] in _FACE_LISTS:
                    corners = [int(x) for x in np.frombuffer(raw, idx_t)]
            else:
                t = np.dtype(order + _ply_type(p[0], path))
                if len(fh.read(t.itemsize)) != t.itemsize:
                    raise ValueError(f"{path}: element '{name}' is truncated in row {k}")
        if name == "face":
            if corners is None:
                raise ValueError(f"{path}: the face element has no vertex_indices list")
            if len(corners) < 3:
                raise ValueError(f"{path}: face {k} has {len(corners)} corners")
            out.append(corners)
    return out


def _check_indices(path, faces, n_vertices):
    if faces.size and (faces.min() < 0 or faces.max() >= n_vertices):
        bad = int(np.flatnonzero((faces < 0).any(axis=1) | (faces >= n_vertices).any(axis=1))[0])
        raise ValueError(f"{path}: triangle {bad} refers to vertex {faces[bad].tolist()} of {n_vertices}")


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
        for lineno, line in enumerate(fh, 1):
            tok = line.split("#", 1)[0].split()
            if not tok:
                continue
            if tok[0] == "v":
                try:
                    verts.append([float(x) for x in tok[1:4]])
                    if len(verts[-1]) != 3:
                        raise ValueError
                except ValueError:
                    raise ValueError(f"{path}:{lineno}: malformed vertex statement") from None
            elif tok[0] == "f":
                if len(tok) < 4:
                    raise ValueError(f"{path}:{lineno}: face with {len(tok) - 1} corners")
                corners = []
                for t in tok[1:]:
                    try:
                        i = int(t.split("/")[0])
                    except ValueError:
                        raise ValueError(f"{path}:{lineno}: malformed face corner '{t}'") from None
                    # 1-based; negative = relative to the vertices read so far (-1: the last one)
                    i = i - 1 if i > 0 else len(verts) + i
                    if i < 0 or i >= len(verts) or t.split("/")[0] == "0":
                        raise ValueError(f"{path}:{lineno}: face corner '{t}' refers to a vertex that does not exist (yet)")
                    corners.append(i)
                faces.extend(_fan([corners]))
    return np.array(verts, dtype=np.float64).reshape(-1, 3), np.array(faces, dtype=np.int32).reshape(-1, 3)


def write_obj(path, vertices, faces):
    with open(path, "w") as fh:
        for p in np.asarray(vertices, dtype=np.float64):
            fh.write("v %r %r %r\n" % tuple(float(x) for x in p))
        for t in np.asarray(faces):
            fh.write("f %d %d %d\n" % tuple(int(x) + 1 for x in t))
