"""The driver parses the LAST stdout line of bench.py and keeps only a bounded tail of the output: a 25 KB line made
BENCH_r05.parsed null. bench.emit() therefore prints every sub-record first and a final headline line
under 4 KB. Checked here with a full-sized set of sub-records (the committed round-5 line, profiles/r05_final_bench.json)."""
import contextlib
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _round5_line():
    return json.load(open(os.path.join(ROOT, "profiles", "r05_final_bench.json")))


def test_final_line_is_small_last_and_complete(tmp_path):
    import bench
    r5 = _round5_line()
    sub = r5.pop("sub_records")
    assert len(json.dumps(sub)) > 12000           # the real thing: what broke the driver's parser
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.emit(r5, {"scaling": "strong", "equals_oracle_golden": True, "gcups": 1.0}, None, sub, str(tmp_path))
    lines = buf.getvalue().rstrip("\n").split("\n")
    final = lines[-1]
    assert len(final.encode()) < 4096
    head = json.loads(final)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "equals_oracle_golden"):
        assert key in head, key
    assert "workload" in head["config"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms"):
        assert key in head["roofline"], key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in head["cpu_baseline"], key
    # every sub-record: its own earlier line, the file, and a golden count in the final line
    names = [json.loads(l)["sub_record"] for l in lines[:-1]]
    assert names == ["strong_scaling"] + list(sub)
    full = json.load(open(tmp_path / "bench_full_record.json"))
    assert full["sub_records"]["configs[3]"] == sub["configs[3]"]
    summ = head["sub_records"]["summary"]
    assert set(summ) == set(names)
    ok, n = summ["band_modes"]["golden"]
    assert ok == n == 16
    assert all(v["golden"][0] == v["golden"][1] for v in summ.values())


def test_oversized_optional_members_are_shed_not_the_contract(tmp_path):
    import bench
    r5 = _round5_line()
    r5.pop("sub_records")
    r5["timed_region"] = "x" * 5000
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        line = bench.emit(r5, None, None, {}, str(tmp_path))
    assert len(line.encode()) < 4096
    head = json.loads(line)
    assert "timed_region" not in head and "roofline" in head and "cpu_baseline" in head and head["value"] == r5["value"]
