"""bench.py's roofline arithmetic (CPU): which resource binds a traversal launch is decided from a committed PMC row and a
launch time -- four fractions, each <= 1 against its own peak, `bound` the largest -- and the committed PMC summaries of this round
must belong to the kernel sources in the tree (bench.py quotes them only while their hash matches)."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_mesh_roofline_names_the_largest_fraction():
    import bench
    # a launch bound by vector issue: 2.4e9 wave-instructions in 5 ms on 1024 SIMDs = 0.39 of the issue peak, little memory traffic
    row = {"SQ_INSTS_VALU": 2.4e9, "thread_cycles_per_valu_inst": 40.0, "wait_share": 0.5, "TCC_HIT_sum": 8e6, "TCC_MISS_sum": 2e6,
           "l2_hit_rate": 0.8, "FETCH_SIZE": 1e5, "WRITE_SIZE": 5e4}
    r = bench.mesh_roofline(row, 5.0, {"bytes_per_launch": 4e9}, 4 << 20)
    assert r["bound"] == "valu" and r["bound_fraction"] == "valu_issue"
    assert abs(r["frac"] - 2.4e9 / 5e-3 / 1e9 / bench.VALU_PEAK_GINST) < 1e-3
    assert set(r["fractions"]) == {"valu_issue", "l2", "beyond_l2", "fabric"}
    assert all(0 <= f["frac"] <= 1 for f in r["fractions"].values())
    assert r["fractions"]["beyond_l2"]["peak"] == bench.IC_GATHER_PEAK_GBS          # a 4 MB tree is served by the Infinity Cache at worst
    assert abs(r["fractions"]["fabric"]["x2_frac"] - (2 * 1e5 + 5e4) * 1024 / 5e-3 / 1e9 / 8000.0) < 1e-3
    assert r["own_bytes"]["gbs"] == 800.0 and r["own_bytes"]["l2_line_bytes_per_own_byte"] == round(1e7 * 128 / 4e9, 3)
    # the same launch with 30 GB of 128-byte fabric reads: the request-size mix decides the fabric fraction, and the fabric binds
    row2 = dict(row, FETCH_SIZE=15e9 / 1024, fabric_read_bytes_by_request_size=30e9)
    r2 = bench.mesh_roofline(row2, 5.0, None, 300 << 20)
    assert r2["bound"] == "hbm" and r2["bound_fraction"] == "fabric"
    assert r2["fractions"]["fabric"]["decided_by"].startswith("request-size") and abs(r2["frac"] - (30e9 + 5e4 * 1024) / 5e-3 / 1e9 / 8000.0) < 1e-3
    assert r2["fractions"]["beyond_l2"]["peak"] == bench.HBM_ACHIEVABLE_GBS         # a tree beyond the Infinity Cache: HBM's rate
    assert bench.mesh_roofline(None, 5.0, None, 1) is None and bench.mesh_roofline(row, 0.0, None, 1) is None


def test_committed_pmc_summaries_belong_to_the_kernel_sources_in_the_tree():
    """`static_pmc` drops a summary whose `sources_sha` is not the hash of clive2_amd/csrc/*: the bench line would then carry no
    fractions.  The newest round's summaries must match, for every workload of the default line."""
    import pytest
    import bench
    sha = bench.kernel_sources_sha()
    newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_cornell.json")))[-1]
    tag = os.path.basename(newest).split("_")[0]
    want = {("cornell", 1920, 1080, 1), ("glass", 1920, 1080, 8), ("blob", 1920, 1080, 8), ("interior", 1920, 1080, 8), ("interior", 3840, 2160, 2),
            ("glass", 1920, 1080, 1), ("blob", 1920, 1080, 1), ("interior", 1920, 1080, 1), ("interior", 3840, 2160, 1)}      # ... and the legs' one_stream forms
    have = set()
    for f in glob.glob(os.path.join(ROOT, "profiles", f"{tag}_pmc_*.json")):
        d = json.load(open(f))
        if d["sources_sha"] != sha:
            # kernels under development: the summaries are stale by design until tools/profile_all.sh has run on the new sources
            pytest.skip(f"{os.path.basename(f)} was profiled on other kernel sources ({d['sources_sha']} != {sha}): re-run tools/profile_all.sh before the round ends")
        have.add((d["scene"], d["width"], d["height"], d["sample_streams"]))
    assert want <= have, want - have
    for scene, w, h, k in want:
        row = bench.static_pmc(scene, w, h, k)
        assert row and row.get("SQ_INSTS_VALU"), (scene, w, h, k)
        if scene != "cornell":
            assert ",false" in row["name"] and row["subpath_kernel"] and ",false" in row["subpath_kernel"]["name"]      # never the tallying variant
            assert row.get("fabric_read_bytes_by_request_size")
