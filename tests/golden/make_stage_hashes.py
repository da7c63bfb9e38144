"""Writes tests/golden/oracle_stage_hashes.json: SHA-256 of the oracle's Path[] buffers, filter aggregators and RNG
buffer after ONE seeded sample of three 64x48 scenes (the Cornell box; the Cornell box with a rough-glass icosphere:
GGX sampling, reflection / transmission, every detmath function; three spheres that carry the material types the shipped
table never reaches).

Why: GPU == oracle proves the two agree with EACH OTHER.  Their elementary functions are sibling files
(oracle/detmath.h, clive2_amd/csrc/detmath.hpp), and an edit applied to both would keep every parity test green while
changing every picture.  The committed hashes pin the arithmetic itself: tests/test_oracle_hardening.py requires the
oracle (CPU) and the HIP path (GPU) to reproduce them.  Re-run this script ONLY for a deliberate change of the pinned
arithmetic, and say so in the commit:   python tests/golden/make_stage_hashes.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "4")


def scenes():
    import numpy as np
    import clive2_amd as c2
    from clive2_amd.load import get_materials
    from clive2_amd.meshes import icosphere
    mats = get_materials()
    mats["alpha"][5] = 0.1
    v, f = icosphere(2, radius=2.0, center=(0.0, 1.0, 0.0))
    # every material type (the shipped table never reaches types 2 and 3, nor type 1 with alpha 0): type 1 smooth,
    # type 2 (Fresnel-weighted reflect / diffuse, alpha 0.3), type 3 (always reflect, alpha 0.05), one sphere each
    from clive2_amd import struct_types as st
    many = np.zeros(10, dtype=st.Material)
    many[:8] = get_materials()
    many[8], many[9] = many[5], many[5]
    many["type"][8], many["alpha"][8] = 2, 0.3
    many["type"][9], many["alpha"][9] = 3, 0.05
    many["color"][9, :3] = (0.9, 0.9, 0.9)
    specs = [dict(mesh=icosphere(2, radius=1.6, center=(x, 0.0, z)), material=m) for x, z, m in ((-3.5, 0.0, 5), (0.0, -1.0, 8), (3.5, 0.0, 9))]
    return {"cornell_64x48": c2.create_scene_from_preset("empty", 64, 48),
            "glass_64x48": c2.create_scene(64, 48, np.array([0, 1.5, 6]), np.array([0, 0, -1]),
                                           file_specs=[dict(mesh=(v, f), material=5)], materials=mats),
            "all_material_types_64x48": c2.create_scene(64, 48, np.array([0, 1.5, 6]), np.array([0, 0, -1]), file_specs=specs, materials=many)}


def oracle_hashes(scene, seed=77):
    from oracle import oracle as orc
    orc.build()
    o = orc.OracleRenderer(scene, seeds=orc.make_seeds(scene.pixel_width * scene.pixel_height, seed=seed))
    o.make_light_rays(); o.make_camera_rays(); o.trace_light_rays(); o.trace_camera_rays(); o.join_paths()
    return orc.stage_hashes(o)


if __name__ == "__main__":
    out = {name: oracle_hashes(s) for name, s in scenes().items()}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_stage_hashes.json")
    json.dump({"seed": 77, "stages": "make_light_rays, make_camera_rays, trace_light_rays, trace_camera_rays, join_paths",
               "scenes": out}, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1))
