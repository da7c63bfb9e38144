"""ABI contract: record layouts vs the golden layout captured from the reference
(src/struct_types.py:4-85) and vs the C structs of the oracle."""
import os
import numpy as np

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_layout_matches_reference_fixture():
    from clive2_amd import struct_types as st
    g = np.load(os.path.join(GOLD, "struct_layout.npz"))
    for name, dt in st.ALL.items():
        assert dt.itemsize == int(g[name]), name
        for field in dt.names:
            assert dt.fields[field][1] == int(g[f"{name}.{field}"]), (name, field)
    # documented sizes
    assert [st.ALL[n].itemsize for n in ("Ray", "Path", "Box", "Triangle", "Material", "Camera")] == \
        [128, 1040, 48, 128, 48, 112]


def test_oracle_structs_match(oracle_mod):
    from clive2_amd import struct_types as st
    L = oracle_mod.lib()
    for code, name in enumerate(("Ray", "Path", "Box", "Triangle", "Material", "Camera")):
        assert L.orc_sizeof(code) == st.ALL[name].itemsize
    assert L.orc_sizeof(6) == st.WeightAggregator.itemsize == 128
    assert oracle_mod.Ray.itemsize == 128 and oracle_mod.Path.itemsize == 1040
    for f in st.Ray.names:
        assert st.Ray.fields[f][1] == oracle_mod.Ray.fields[f][1]
    assert st.WeightAggregator.fields["total_contribution"][1] == 48
    assert st.WeightAggregator.fields["contrib_weight_sum"][1] == 64
