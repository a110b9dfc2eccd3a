"""CPU-only: the line bench.py prints for the driver and the versioning of the counter records behind its rooflines.

Round 4's line was 30 KB; the driver keeps an 8 KB tail of stdout, so nothing of it could be parsed.  The short line is built
from the full record by `bench.short_line`; here it is built from a recorded full record (round 4's, plus every block this
round adds) and held under 4 KB.  The counter records (`profiles/pmc_<workload>.json`) carry the ISA hash of every kernel
they were counted on; a record of another build must be detected and left unused.
"""
import copy
import json
import os

import pytest

import bench
from mxx_amd import codeobj

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def recorded_full_line():
    with open(os.path.join(ROOT, "profiles", "r04_bench_default.json")) as f:
        full = json.load(f)
    # the blocks added after that record, shaped as bench.py builds them
    m4 = copy.deepcopy(full["chain_m4"])
    m4.update({"requests_in_flight": 16, "speedup_vs_one_request": 5.123456, "value": 36543.21})
    full["chain_m4_batched"] = m4
    ref = copy.deepcopy(full["preimage"])
    ref["vs_extension_sequence"] = 1.3812345
    full["preimage_reference_sequence"] = ref
    full["compact_bytes"] = {"ms_per_step": 4.9123, "value": 1.234e10, "unit": "B/s", "roofline": {"frac": 0.4123}, "payload_bytes": 61234567,
                             "kernel_ms": 1.2345, "d2h_ms": 2.3456, "pcie_GBps": 26.1234, "host_ms": 0.1234}
    full["gate_batch"] = {"ms_per_step": 0.082123, "value": 195123.4, "unit": "products/s", "roofline": {"frac": 0.2123},
                          "loop_ms": 0.463123, "speedup_vs_loop": 5.64123}
    full["preimage_mixed_keys"] = {"ms_per_step": 2.1234, "value": 30123.4, "unit": "preimages/s", "roofline": {"frac": None},
                                   "requests_in_flight": 16, "loop_ms": 4.81234, "speedup_vs_loop": 2.2661}
    full["sustained"] = {"steps": 8621, "seconds": 5.0123456, "ms_per_step": 0.58141234, "value": 6191234.5}
    full["roofline"]["counters_stale"] = False
    for blk in (full["preimage"], full["preimage_m3b"], full["chain_m4"], full["chain_m4_batched"]):
        blk["roofline"].update({"frac_useful": 0.3123, "lane_utilisation": 0.4212, "counters_stale": False})
    return full


def test_short_line_fits_the_drivers_tail_and_keeps_the_contract_keys():
    full = recorded_full_line()
    assert len(json.dumps(full)) > 20000  # the record that broke round 4's parse
    line = bench.short_line(full)
    text = json.dumps(line)
    assert len(text) < 4096, len(text)
    assert "\n" not in text
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["metric"] == "dcrt_ring_ops_per_s" and line["unit"] == "ring-ops/s" and line["vs_baseline"] is None
    assert line["config"]["workload"].startswith("M2A") and "model" not in line["config"]
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-3)
    assert rf["traffic"] and rf["algorithmic_bytes_per_launch"] and rf["kernel_ms"]
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 16 and cb["value"] > 0 and cb["sample"]
    assert line["value"] == pytest.approx(full["value"], rel=1e-6) and line["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    # one record per BASELINE configuration (+ the three the round-4 review asked for, + the batched chain)
    assert set(line["configs"]) == {"m1_ntt_mul", "m2b_product", "m2b_decompose", "m2b_mul_decompose", "m3a_preimage",
                                    "m3a_reference_sequence", "m3b_preimage", "m4_chain", "m4_chain_batched", "m4_mixed_keys",
                                    "compact_bytes", "gate_batch"}
    for name, rec in line["configs"].items():
        assert rec["ms_per_step"] and rec["value"] and rec["unit"], name
    assert line["configs"]["m3a_preimage"]["frac_useful"] == 0.3123
    assert line["configs"]["m4_chain_batched"]["requests"] == 16 and "sustained" in line and "detail" in line
    assert "counters_stale" not in line["configs"]["m3a_preimage"]  # reported only when true
    assert line["configs"]["m1_ntt_mul"]["ntt_frac"] == full["kernels"]["ntt_forward"]["frac_of_hbm_peak"]


def test_short_line_sheds_detail_rather_than_outgrowing_the_limit():
    full = recorded_full_line()
    full["config"]["workload"] = "M2A " + "x" * 4000
    full["config"]["sharding"] = "y" * 4000
    for i in range(40):  # far more blocks than any run produces
        full[f"extra{i}"] = full["chain_m4"]
    line = bench.short_line(full)
    assert len(json.dumps(line)) <= bench.SHORT_LINE_LIMIT
    assert line["roofline"]["frac"] and line["cpu_baseline"]["value"]


def test_short_line_at_n_gpus_carries_the_exchange_verdict():
    full = recorded_full_line()
    full["n_gpus"] = 2
    full["cpu_baseline"] = None
    full["exchange"] = {"ranks_seen": 2, "comm_backend": "torch.distributed 'nccl' (RCCL)", "devices": [0, 1], "peer_access": [[1, 1], [1, 1]],
                        "foreign_blocks_checked_per_rank": [1, 1], "self_validated": True, "distinct_devices": True,
                        "comm_verified_on_distinct_devices_before_this_run": False, "comm_verified_by_this_run": True}
    full["preimage"]["exchange"] = dict(full["exchange"])
    full["gather"] = "step"
    full["other_gather"] = {"gather": "lazy", "ms_per_step": 0.0912345, "value": 39458123.4}
    full["independent_units"] = {"scaling": "weak", "value": 12345678.9, "ms_per_step": 0.5912345, "units_per_step": 7200, "sharding": "x"}
    line = bench.short_line(full)
    assert line["gather"] == "step" and line["other_gather"] == {"gather": "lazy", "value": 39458000, "ms_per_step": 0.091234}
    assert line["independent_units"]["scaling"] == "weak" and line["independent_units"]["value"] == 12346000
    assert line["exchange"] == {"ranks_seen": 2, "comm_backend": "torch.distributed 'nccl' (RCCL)", "self_validated": True,
                                "distinct_devices": True, "comm_verified_by_this_run": True, "preimage_self_validated": True}
    assert line["cpu_baseline"] is None
    assert len(json.dumps(line)) < 4096


def test_dry_run_objects_pass_through():
    assert bench.short_line({"dry_run": True, "world": 2}) == {"dry_run": True, "world": 2}


def test_sig_rounds_to_significant_digits():
    assert bench.sig(6134688.61730078, 7) == 6134689.0
    assert bench.sig(0.5868268504855223, 6) == 0.586827
    assert bench.sig(13.509383658328908) == 13.509
    assert bench.sig(None) is None and bench.sig(26) == 26 and bench.sig(True) is True and bench.sig(0.0) == 0.0


# ---------------------------------------------------------------------------------------------------
# versioned counter records
# ---------------------------------------------------------------------------------------------------
def test_library_kernels_have_isa_hashes():
    h = codeobj.kernel_isa_hashes()
    for k in ("matmul_kernel", "mmdma32::kernel_u32", "ntt14::fwd_kernel", "ntt14::inv_kernel", "ntt14::fwd_digits_kernel",
              "gauss_samp_lanes_kernel", "sample_gauss_kernel", "p1_sample_lanes_kernel", "elementwise_kernel", "*"):
        assert k in h and len(h[k]) == 16, k
    assert codeobj.kernel_isa_hashes() is h  # cached per (path, mtime)
    assert codeobj.kernel_base("void ntt14::fwd_kernel<unsigned int, false>(unsigned int*, int)") == "ntt14::fwd_kernel"
    assert codeobj.kernel_base("(ntt14::fwd_kernel<W, TIGHT>)") == "ntt14::fwd_kernel"


def test_a_counter_record_of_another_build_is_detected(tmp_path, monkeypatch):
    cur = codeobj.kernel_isa_hashes()
    rec = {"workload": "m2a", "steps": 3, "head": "abc1234", "gpupoly_version": "x", "source": "test",
           "kernels": {"matmul_kernel": {"launches_per_step": 1.0, "isa_hash": cur["matmul_kernel"], "hbm_bytes_per_launch": 3.69e9,
                                         "SQ_INSTS_VALU": 3.0e7},
                       "ntt14::fwd_kernel": {"launches_per_step": 1.0, "isa_hash": "0123456789abcdef", "hbm_bytes_per_launch": 1.0,
                                             "SQ_INSTS_VALU": 1.0},
                       "elementwise_kernel": {"launches_per_step": 1.0, "hbm_bytes_per_launch": 1.0, "SQ_INSTS_VALU": 1.0}}}
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "pmc_m2a.json").write_text(json.dumps(rec))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    got = bench.load_counters("m2a")
    assert got["file"] == "profiles/pmc_m2a.json" and got["head"] == "abc1234"
    assert got["kernels"]["matmul_kernel"]["stale"] is False
    assert got["kernels"]["ntt14::fwd_kernel"]["stale"] is True       # counted on other ISA text
    assert got["kernels"]["elementwise_kernel"]["stale"] is True      # no hash at all (a round-4 record)
    assert got["stale_kernels"] == ["elementwise_kernel", "ntt14::fwd_kernel"]
    assert bench.load_counters("m3a") is None
    assert codeobj.stale_kernels({"matmul_kernel": cur["matmul_kernel"], "ntt14::fwd_kernel": "0" * 16}) == ["ntt14::fwd_kernel"]
    assert codeobj.stale_kernels(None) == ["*"]


def test_round4_records_without_hashes_count_as_stale():
    rec = bench.load_counters("m2a", current_hashes=codeobj.kernel_isa_hashes())
    if rec is None or rec["file"].endswith("/pmc_m2a.json"):
        pytest.skip("a versioned record exists; the round-4 fallback is not in use")
    assert rec["file"] == "profiles/r04_pmc_m2a.json" and rec["kernels"]["matmul_kernel"]["stale"] is True


def test_committed_counter_records_match_the_library_or_are_flagged():
    """every versioned record under profiles/ either matches today's kernels or bench.py will say `counters_stale`; the
    valu-mix prices follow the same rule (entries of other builds are dropped)"""
    cur = codeobj.kernel_isa_hashes()
    for wl in ("m1", "m2a", "m2b", "m2b_decompose", "m2b_mul_decompose", "m3a", "m3b", "m4"):
        rec = bench.load_counters(wl, current_hashes=cur)
        if rec is None:
            continue
        for base, k in rec["kernels"].items():
            if base.startswith("__amd_rocclr"):
                continue  # the HIP runtime's copy kernels
            assert k["stale"] == (k.get("isa_hash") is None or k["isa_hash"] != cur.get(base))
    mix = bench.load_valu_mix(cur)
    for base, k in mix.items():
        assert k["isa_hash"] == cur[base] and 2.5 <= k["cycles_per_inst"] <= 4.5
