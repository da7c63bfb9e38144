"""Procedural stand-in meshes for the configs whose assets are not in the reference snapshot
(SURVEY.md F10, §8d C3-C5): icospheres, a noisy "bunny-sized" blob, an instanced interior."""
import numpy as np

_T = (1.0 + 5.0 ** 0.5) / 2.0
_ICO_V = [(-1, _T, 0), (1, _T, 0), (-1, -_T, 0), (1, -_T, 0), (0, -1, _T), (0, 1, _T),
          (0, -1, -_T), (0, 1, -_T), (_T, 0, -1), (_T, 0, 1), (-_T, 0, -1), (-_T, 0, 1)]
_ICO_F = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4),
          (11, 10, 2), (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8),
          (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]


def icosphere(subdiv, radius=1.0, center=(0.0, 0.0, 0.0)):
    """Unit icosahedron subdivided `subdiv` times (20*4^subdiv faces), outward winding."""
    v = np.array(_ICO_V, dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array(_ICO_F, dtype=np.int64)
    for _ in range(subdiv):
        edges = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), axis=1)
        uniq, inv = np.unique(edges, axis=0, return_inverse=True)
        inv = inv.reshape(-1)
        mids = v[uniq[:, 0]] + v[uniq[:, 1]]
        mids /= np.linalg.norm(mids, axis=1, keepdims=True)
        base = len(v)
        v = np.concatenate([v, mids])
        m = len(f)
        ab, bc, ca = base + inv[:m], base + inv[m:2 * m], base + inv[2 * m:]
        a, b, c = f[:, 0], f[:, 1], f[:, 2]
        f = np.concatenate([np.stack([a, ab, ca], 1), np.stack([b, bc, ab], 1),
                            np.stack([c, ca, bc], 1), np.stack([ab, bc, ca], 1)])
    return v * radius + np.asarray(center, dtype=np.float64), f.astype(np.int32)


def noisy_blob(subdiv=6, radius=2.5, center=(0.0, 1.0, 0.0), amplitude=0.08, seed=20240928):
    """C4 stand-in: icosphere with seeded low-frequency radial noise (81,920 tris at subdiv 6)."""
    v, f = icosphere(subdiv)
    rng = np.random.RandomState(seed)
    freq = rng.normal(size=(6, 3)) * 3.0
    phase = rng.uniform(0, 2 * np.pi, size=6)
    bump = np.sin(v @ freq.T + phase).sum(axis=1) / 6.0
    return v * (radius * (1.0 + amplitude * bump))[:, None] + np.asarray(center), f


def interior_grid(n=7, subdiv=5, radius=1.2, y=-0.8, extent=8.4):
    """C5 stand-in: n x n icospheres resting on the Cornell floor (1,003,520 tris at 7x7, subdiv 5).
    Returns a list of (vertices, faces, material) with materials alternating 4 / 5."""
    sv, sf = icosphere(subdiv, radius=radius)
    xs = np.linspace(-extent, extent, n)
    out = []
    for i, x in enumerate(xs):
        for j, z in enumerate(xs):
            out.append((sv + np.array([x, y, z]), sf, 4 if (i + j) % 2 == 0 else 5))
    return out
