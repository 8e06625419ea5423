"""The stdout line of bench.py must survive the driver's capture (BENCH_r04.json: "parsed": null -- the line had grown to 22.8 KB).
bench.compact_line builds the line from the full record; the full record goes to profiles/bench_full.json and stderr."""
import contextlib
import copy
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402  (imports nothing heavy at module level)

CANNED = os.path.join(ROOT, "tests", "golden", "bench_full_r04.json")   # the full record of round 4's closing run (22.8 KB)
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def canned():
    return json.load(open(CANNED))


def test_line_from_the_round_4_record_is_short_and_complete():
    full = canned()
    assert len(json.dumps(full)) > 20000
    line = bench.compact_line(full)
    s = json.dumps(line)
    assert len(s) < bench.LINE_LIMIT == 4096
    back = json.loads(s)
    for k in CONTRACT:
        assert k in back, k
    assert back["value"] == float(f"{full['value']:.5g}")
    assert "model" not in back["config"] and back["config"]["workload"].startswith("BFV N=2^14")
    roof = back["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "essential_frac", "hbm_frac", "ctmul_hbm_frac", "kernel",
              "algorithmic_bytes", "traffic_over_algorithmic", "pmc_source_id", "pmc_stale"):
        assert k in roof, k
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    cb = back["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 128 and cb["value"] > 0 and cb["single_thread"]["value"] > 0 and cb["sample"]
    assert set(back["ntt"]) >= {"fwd_GBs", "inv_GBs", "fwd_frac", "inv_frac"}
    oc = back["other_configs"]
    assert len(oc) == len(full["other_configs"])
    assert len({c["name"] for c in oc}) == len(oc)                       # both MNIST circuit shapes stay distinguishable
    cfg3 = next(c for c in oc if c["name"].startswith("cfg#3"))
    assert cfg3["ks_s"] > 0 and cfg3["rot_s"] > 0 and cfg3["kern"]["keyswitch"][0].startswith("k_ks_fused_sub")
    assert back["full_record"] == "profiles/bench_full.json"


def test_line_stays_below_the_limit_whatever_the_record_holds():
    full = canned()
    # an 8-rank run: per-rank arrays, long error strings, more configurations than any run has
    multi = []
    for i in range(6):
        multi.append({"config": f"cfg#{i} " + "x" * 200, "unit": "key switches/s", "value": 1.0e6 + i, "ms_per_step": 3.14159, "scaling": "strong",
                      "global_units": 4096, "nranks_seen": 8, "imbalance_max_over_min": 1.01234567,
                      "per_rank": [{"rank": r, "units": 512, "s": 0.1, "units_per_s": 5120.0} for r in range(8)],
                      "gather": {"collective": "y" * 300, "ms": 1.0}, "note": "z" * 2000})
    full["configs_multi"] = multi
    full["other_configs"] = full["other_configs"] * 3
    full["errors"] = ["e" * 5000] * 10
    full["gather"] = {"collective": "c" * 1000, "ms_per_step": 1.0, "GBs_per_rank": 100.0, "value_with_gather": 5.0e5, "fallback": "f" * 1000}
    line = bench.compact_line(full)
    s = json.dumps(line)
    assert len(s) < 4096
    back = json.loads(s)
    for k in CONTRACT + ("roofline", "cpu_baseline"):
        assert k in back, k                                              # what is dropped to fit is never the contract


def test_line_with_oversized_unbounded_parts_is_forced_to_fit():
    """config extras, a plain roofline dict and the cpu_baseline survive every ordinary shrink step; the fit is enforced
    after them -- first their extras go, last of all everything but the contract keys."""
    full = canned()
    full["config"] = dict(full["config"], **{f"extra_{i}": "k" * 300 for i in range(40)})
    line = bench.compact_line(full)
    s = json.dumps(line)
    assert len(s) < 4096
    back = json.loads(s)
    for k in CONTRACT:
        assert k in back, k
    assert back.get("truncated") is True and "workload" in back["config"]
    # a roofline of many scalar fields (the non-headline form) alone: its six contract fields stay
    full = canned()
    full["roofline"] = dict({"bound": "hbm", "achieved": 1.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.1, "traffic": None},
                            **{f"n{i}": "r" * 79 for i in range(80)})
    back = json.loads(json.dumps(bench.compact_line(full)))
    assert len(json.dumps(back)) < 4096 and back["roofline"]["frac"] == 0.1 and "n0" not in back["roofline"]
    for k in CONTRACT + ("roofline",):
        assert k in back, k


def test_emit_prints_the_compact_line_last_on_stdout_and_the_full_record_elsewhere(tmp_path, monkeypatch):
    full = canned()
    monkeypatch.setattr(bench, "FULL_RECORD", str(tmp_path / "bench_full.json"))
    out, err = io.StringIO(), io.StringIO()
    with contextlib.redirect_stdout(out), contextlib.redirect_stderr(err):
        bench.emit(copy.deepcopy(full))
    lines = [l for l in out.getvalue().splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[-1]) < 4096
    assert json.loads(lines[-1])["value"] > 0
    assert json.load(open(tmp_path / "bench_full.json")) == full          # every digit, every note
    assert json.loads(err.getvalue().strip().splitlines()[-1]) == full


def test_line_of_a_multi_rank_configuration_run():
    rec = {"metric": "key switches/s -- cfg#4", "value": 2.0e6, "unit": "key switches/s", "n_gpus": 8, "steps": 5, "warmup": 2, "ms_per_step": 2.0,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
           "config": {"workload": "cfg#4 BGV N=2^14 L=6 keyswitch", "global_units": 4096, "sharding": "units x8, no data-path collective"},
           "roofline": {"bound": "hbm", "achieved": 400.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.05, "traffic": None, "note": "n" * 500},
           "per_rank": [{"rank": r, "units": 512, "s": 0.01, "units_per_s": 2.5e5} for r in range(8)], "nranks_seen": 8,
           "imbalance_max_over_min": 1.02, "cabi_gather_error": "RuntimeError: tfhe_comm_create: rendezvous deadline " + "d" * 400,
           "gather": {"collective": "all_gather_into_tensor (RCCL over xGMI) -- fallback", "ms": 1.5, "GBs_per_rank": 90.0, "value_with_gather": 1.5e6}}
    line = bench.compact_line(rec)
    s = json.dumps(line)
    assert len(s) < 4096
    assert line["nranks_seen"] == 8 and line["roofline"]["bound"] == "hbm" and line["roofline"]["frac"] == 0.05
    assert line["gather"]["fallback"].startswith("RuntimeError: tfhe_comm_create") and len(line["gather"]["fallback"]) <= 120
    assert "per_rank" not in line
