"""bench.py's stdout contract, without a GPU: the one line the driver parses is built by bench.compact_line from the full
result and must stay under 4 KB whatever the result holds (round 5 printed the full 28 KB object and the driver's record
of the run came back with `parsed: null`)."""
import importlib.util
import json
import os

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _full_results():
    """Full results of real runs: the lines round 5 printed (profiles/r05_bench_lines.jsonl keeps them whole)."""
    rows = []
    with open(os.path.join(ROOT, "profiles", "r05_bench_lines.jsonl")) as f:
        for l in f:
            if l.startswith("{"):
                d = json.loads(l)
                rows.append(d.get("bench", d))
    return [r for r in rows if "roofline" in r and "paths" in r]


def test_line_is_short_and_carries_the_contract():
    b = _bench()
    fulls = _full_results()
    assert fulls and max(len(json.dumps(f)) for f in fulls) > 20000           # what the driver could not parse
    for full in fulls:
        txt = b.compact_line(full)
        assert "\n" not in txt and len(txt) < 4096
        line = json.loads(txt)
        for key in ("metric", "unit", "n_gpus", "steps", "warmup", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
            assert line[key] == full[key]
        assert abs(line["value"] - full["value"]) <= 1e-6 * full["value"]
        assert abs(line["ms_per_step"] - full["ms_per_step"]) <= 1e-6 * full["ms_per_step"]
        r, fr = line["roofline"], full["roofline"]
        assert r["bound"] == fr["bound"] and r["unit"] == fr["unit"] and r["peak"] == fr["peak"] and r["kernel"] == fr["kernel"]
        assert abs(r["frac"] - fr["frac"]) < 1e-5 and abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-4
        assert abs(r["traffic"] - fr["traffic"]) <= 1e-5 * fr["traffic"]
        c = line["cpu_baseline"]
        assert c["kind"] == "port" and c["cores"] == full["cpu_baseline"]["cores"] and c["unit"] == full["unit"] and len(c["sample"]) <= 160
        assert len(line["paths_summary"]) == len(full["paths"]) + 1 and "dropped_for_size" not in line
        assert "model" not in line["config"] and len(line["config"]["workload"]) < 128


def test_line_stays_short_when_the_result_grows():
    b = _bench()
    full = json.loads(json.dumps(_full_results()[-1]))
    for i in range(400):                                            # far more rows than the hot path has
        row = json.loads(json.dumps(full["paths"][0]))
        row["path"] = "an extra row of the hot path, number %d, with a long name" % i
        full["paths"].append(row)
    full["config"]["workload"] = "x" * 5000
    txt = b.compact_line(full)
    line = json.loads(txt)
    assert len(txt) < 4096 and "paths_summary" in line["dropped_for_size"]
    assert line["roofline"]["frac"] > 0 and line["cpu_baseline"]["value"] > 0 and line["value"] > 0


def test_path_ids_are_distinct():
    b = _bench()
    names = [p["path"] for p in _full_results()[-1]["paths"]]
    ids = [b.path_id(n) for n in names]
    assert len(set(ids)) == len(ids), ids


def test_detail_file_roundtrip(tmp_path):
    b = _bench()
    full = _full_results()[-1]
    p = str(tmp_path / "d.json")
    b.write_detail(full, p)
    assert json.load(open(p)) == full
