"""How the three binary PLY fixtures of this directory were packed: `struct` with an explicit byte order, NOT the package's
own writer (a reader tested against its own writer proves little).  The ASCII .ply / .obj files were typed by hand."""
import struct

verts = [(0.0, 0.0, 0.0), (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)]
faces = [(0, 2, 1), (0, 1, 3), (1, 2, 3), (0, 3, 2)]
head = (b"ply\nformat binary_big_endian 1.0\ncomment hand-packed fixture (tests/golden/meshes/make_binary_fixtures.py): a tetrahedron, big-endian\n"
        b"element vertex 4\nproperty float x\nproperty float y\nproperty float z\nelement face 4\nproperty list uchar int vertex_indices\nend_header\n")
open("tetra_big_endian.ply", "wb").write(head + b"".join(struct.pack(">fff", *v) for v in verts) + b"".join(struct.pack(">Biii", 3, *f) for f in faces))
head = (b"ply\nformat binary_little_endian 1.0\ncomment hand-packed fixture: a quad among triangles (the row-by-row path), ushort counts, a property behind the list\n"
        b"element vertex 5\nproperty float x\nproperty float y\nproperty float z\nproperty uchar quality\nelement face 3\n"
        b"property list ushort int vertex_indices\nproperty float area\nend_header\n")
v5 = [(0, 0, 0, 1), (2, 0, 0, 2), (2, 2, 0, 3), (0, 2, 0, 4), (1, 1, 3, 5)]
body = b"".join(struct.pack("<fffB", *v) for v in v5)
body += struct.pack("<Hiiii", 4, 0, 1, 2, 3) + struct.pack("<f", 4.0)
body += struct.pack("<Hiii", 3, 0, 1, 4) + struct.pack("<f", 1.5)
body += struct.pack("<Hiii", 3, 1, 2, 4) + struct.pack("<f", 1.5)
open("mixed_little_endian.ply", "wb").write(head + body)
open("truncated_big_endian.ply", "wb").write(open("tetra_big_endian.ply", "rb").read()[:-7])
