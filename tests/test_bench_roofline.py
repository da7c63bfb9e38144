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
    # round 6: the vector L1's look-up rate (TCP_TOTAL_CACHE_ACCESSES against one look-up per clock and CU) -- where it is the largest
    # fraction the launch is bound by it: 2.7e9 look-ups in 5 ms on 256 CUs = 0.879 per clock and CU
    row3 = dict(row, TCP_TOTAL_CACHE_ACCESSES_sum=2.7e9)
    r3 = bench.mesh_roofline(row3, 5.0, None, 4 << 20)
    assert r3["bound"] == "l1" and r3["bound_fraction"] == "l1_lookups" and set(r3["fractions"]) == {"valu_issue", "l1_lookups", "l2", "beyond_l2", "fabric"}
    assert abs(r3["frac"] - 2.7e9 / 5e-3 / (256 * 2.4e9)) < 1e-3 and r3["fractions"]["l1_lookups"]["frac_of_measured_peak"] > r3["frac"]


def test_committed_pmc_summaries_belong_to_the_kernel_sources_in_the_tree():
    """`static_pmc` drops a summary whose `sources_sha` is not the hash of clive2_amd/csrc/*: the bench line would then carry no
    fractions.  The newest round's summaries must match, for every workload of the default line."""
    import pytest
    import bench
    sha = bench.kernel_sources_sha()
    newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_cornell.json")))[-1]
    tag = os.path.basename(newest).split("_")[0]
    want = {("cornell", 1920, 1080, 1), ("glass", 1920, 1080, 16), ("blob", 1920, 1080, 16), ("interior", 1920, 1080, 16), ("interior", 3840, 2160, 2),
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


def _stored_full_result():
    return json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default.json")))


def test_stdout_line_is_one_short_json_line_with_roofline_and_cpu_baseline():
    """BENCH_r05.parsed was null: the line had grown to 22.8 KB and the driver keeps 8,000 characters of stdout.  The line is
    now built by bench.compact_line from the full result (stored here: round 5's own 22.8 KB result) and capped at 4,000."""
    import bench
    full = _stored_full_result()
    assert len(json.dumps(full)) > 20000
    text = bench.compact_line(full)
    assert len(text) <= bench.LINE_CAP == 4000 and "\n" not in text
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"] and line["config"]["workload"] == full["config"]["workload"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "rays_per_launch"):
        assert line["roofline"][k] == full["roofline"][k], k
    assert line["roofline"]["hbm"]["bytes_per_ray"] == full["roofline"]["hbm"]["bytes_per_ray"]
    for k in ("value", "unit", "cores", "kind"):
        assert line["cpu_baseline"][k] == full["cpu_baseline"][k]
    assert set(line["legs"]) == {"c3_glass_5k", "c4_blob_82k", "c5_interior_1m", "c5_interior_1m_4k"}
    for leg in line["legs"].values():
        assert len(json.dumps(leg)) <= 260 and not any(isinstance(v, str) and len(v) > 16 for v in leg.values())      # numbers, no prose
    assert line["legs"]["c5_interior_1m"]["mrays_per_s"] == full["roofline_hbm"]["mrays_per_s"]
    assert line["legs"]["c5_interior_1m"]["k1_mrays_per_s"] == full["roofline_hbm"]["one_stream"]["mrays_per_s"]


def test_stdout_line_keeps_its_cap_whatever_the_result_grows_to():
    """The cap is enforced, not hoped for: an N > 1 result with a communicator object of 16 devices, a strong-scaling leg, failed
    legs with long error texts and a workload string of a kilobyte still gives <= 4,000 characters with `roofline` and
    `cpu_baseline` intact."""
    import bench
    full = _stored_full_result()
    full["n_gpus"] = 8
    full["comm"] = {"nranks": 8, "rank0_device": "0000:05:00.0", "devices": ["0000:%02x:00.0" % i for i in range(16)] * 40, "distinct_devices": 8,
                    "launcher_world_size": 8, "allreduce_ms": 1.234, "allreduce_bytes": 66355200}
    full["strong_scaling"] = {"workload": "x" * 3000, "scaling": "strong", "value": 1.0, "unit": "Mrays/s", "seconds": 1.0, "comm": full["comm"],
                              "samples_rendered_all_ranks": 1024, "ms_per_sample_whole_job": 1.0}
    full["roofline_blob"] = {"error": "RendererError('" + "y" * 5000 + "')", "leg_wall_s": 1.0}
    full["config"]["workload"] = "z" * 1000
    full["roofline"]["static"]["what"] = "w" * 5000
    for i in range(40):
        full[f"future_key_{i}"] = {"text": "q" * 500}
    text = bench.compact_line(full)
    assert len(text) <= 4000
    line = json.loads(text)
    assert line["roofline"]["frac"] == full["roofline"]["frac"] and line["cpu_baseline"]["value"] == full["cpu_baseline"]["value"]
    assert line["legs"]["c4_blob_82k"]["error"].startswith("RendererError") and len(line["legs"]["c4_blob_82k"]["error"]) <= 80
    assert line["comm"]["nranks"] == 8 and len(line["comm"]["devices"]) == 16
    # pathological: every string field huge -> extras are shed, the contract's keys stay
    full["config"]["parallelism"] = "p" * 6000
    line = json.loads(bench.compact_line(full))
    assert "roofline" in line and "cpu_baseline" in line and "value" in line


def test_emit_writes_the_full_result_beside_the_script_and_one_line_to_stdout(tmp_path, monkeypatch, capfd):
    import bench
    full = _stored_full_result()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "gpurun_out")
    bench.emit(full)
    out, err = capfd.readouterr()
    assert out.count("\n") == 1 and len(out) <= 4001 and json.loads(out)["detail"] == "bench_detail.json"
    assert len(err) < 300                                            # a bounded tail of both streams must still hold the line
    for path in (tmp_path / "bench_detail.json", tmp_path / "gpurun_out" / "bench_detail.json"):
        assert json.load(open(path)) == full


def test_pass_fabric_sums_committed_bytes_over_the_kernels_of_a_pass():
    """`pass_fabric` = sum over kernels (committed fabric bytes per launch x launches per pass) / pass time / 8 TB/s.
    Hand-made PMC rows: connection walk 400 GB, eleven subpath-walk launches of which one is the merged level-0 launch (10 x 20 GB
    + 1 x 10 GB), a bounce kernel launched 6 times per pass (24 launches over the 4 passes of the PMC run) at 2 GB, set-up 1 GB;
    the tallying variants of the walks (warm-up only) do not count."""
    import bench
    def row(read_gb, write_gb=0.0, launches=4):
        return {"fabric_read_bytes_by_request_size": read_gb * 1e9, "WRITE_SIZE": write_gb * 1e9 / 1024.0, "FETCH_SIZE": 0.0, "launches_fetch": launches}
    conn, sub = "k_traverse_wide<1,ConnRaySource,false,true,true>", "k_traverse_wide<1,PathRaySource,false,true,true>"
    kernels = {conn: row(390.0, 10.0, 2), sub: row(20.0, 0.0, 20), "k_traverse_wide<1,DualPathRaySource,false,true,true>": row(10.0, 0.0, 2),
               "k_traverse_wide<1,ConnRaySource,true,true,true>": row(999.0, 0.0, 1), "k_traverse_persistent<true,false,ConnRaySource>": row(999.0, 0.0, 1),
               "k_trace_subpath<false,false,true>": row(1.0, 1.0, 24), "k_connect_setup": row(0.5, 0.5, 4)}
    pf = bench.pass_fabric(kernels, conn, sub, 11.0, 100.0)
    want = 400.0 + 10 * 20.0 + 10.0 + 6 * 2.0 + 1.0
    assert abs(pf["bytes_per_pass"] - want * 1e9) < 1e6
    assert abs(pf["gbs"] - want / 0.1) < 1.0 and abs(pf["frac"] - want / 0.1 / 8000.0) < 1e-3
    assert bench.pass_fabric(None, conn, sub, 11.0, 100.0) is None and bench.pass_fabric(kernels, conn, sub, 11.0, 0.0) is None
    # a row without the request-size pass: 2 x FETCH_SIZE + WRITE_SIZE (the guide's correction)
    assert bench.fabric_bytes({"FETCH_SIZE": 1000.0, "WRITE_SIZE": 500.0}) == (2 * 1000.0 + 500.0) * 1024.0
