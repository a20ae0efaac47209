"""bench.py's one-line JSON contract (the driver parses it), on a small database so it runs in seconds;
and the N>1 path of bench.py itself -- MatchPipeline's two-all-gather protocol under torch.distributed --
rehearsed with two gloo ranks sharing the one GPU of the box, against the one-rank run."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


LINE_LIMIT = 4096                                                    # what the driver's parser is known to take


def run_bench(extra, launcher=None, timeout=900, want_line=False):
    """Runs bench.py; checks the stdout contract -- exactly ONE JSON line, the LAST thing on stdout, under 4 KB (round 5's
    28 KB line was not parsed by the driver), with a short stderr -- and returns the full result from the sidecar file
    (and the parsed line with want_line)."""
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        detail = os.path.join(tmp, "detail.json")
        cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + extra + ["--detail", detail]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
        assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-3000:])
        full = json.load(open(detail))
    out_lines = res.stdout.splitlines()
    lines = [l for l in out_lines if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]                       # exactly ONE JSON line
    assert out_lines[-1] == lines[0]                                 # ... and nothing behind it
    assert len(lines[0]) < LINE_LIMIT, len(lines[0])
    assert len(res.stdout) + len(res.stderr) < 2 * LINE_LIMIT, (len(res.stdout), len(res.stderr))   # the driver keeps a tail
    line = json.loads(lines[0])
    for key in ("metric", "unit", "n_gpus", "steps", "warmup", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert line[key] == full[key], key
    assert abs(line["value"] - full["value"]) <= 1e-5 * full["value"]
    assert abs(line["ms_per_step"] - full["ms_per_step"]) <= 1e-5 * full["ms_per_step"]
    assert abs(line["roofline"]["frac"] - full["roofline"]["frac"]) <= 1e-4 * full["roofline"]["frac"]
    assert "dropped_for_size" not in line
    return (full, line) if want_line else full


def test_bench_json_contract():
    d, line = run_bench(["--gpus", "1", "--steps", "4", "--warmup", "1", "--rows", "30000", "--cpu-sample-rows", "30000",
                         "--no-power-probe", "--path-frames", "24"], want_line=True)
    # the line itself: the contract keys, roofline, cpu_baseline, the agreement, one triple per other row of the hot path
    for key, typ in [("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str),
                     ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict), ("paths_summary", dict)]:
        assert isinstance(line[key], typ), key
    assert line["vs_baseline"] is None and "workload" in line["config"] and "model" not in line["config"]
    lr = line["roofline"]
    assert lr["bound"] == "hbm" and lr["unit"] == "GB/s" and lr["peak"] == 8000.0 and lr["kernel"] == "score_gemm_kernel"
    assert abs(lr["frac"] - lr["achieved"] / lr["peak"]) < 1e-4 and "traffic" in lr and lr["kernel_ms"] > 0
    lc = line["cpu_baseline"]
    assert lc["kind"] == "port" and lc["value"] > 0 and lc["cores"] >= 1 and lc["unit"] == line["unit"] and lc["sample"]
    assert line["recall_at_1"] == 1.0 and line["topk_index_agreement_vs_oracle"] == 1.0
    assert len(line["paths_summary"]) == len(d["paths"]) + 1
    assert all(len(v) == 3 and v[0] > 0 and v[1] > 0 for k_, v in line["paths_summary"].items() if k_ != "_")
    sm = line["rccl_world1_smoke"]
    assert sm["ok"] is True and sm["backend"] == "nccl" and sm["world"] == 1 and sm["equals_one_shot"] is True
    assert sm["exhaustive_round_equals_one_shot"] is True and min(sm["allgather_us"]) > 0
    assert set(line["configs_summary"]) == {"_", "cfg3", "cfg4"} and set(line["emulated_ranks_qfps"]) == {"_", "2", "4", "8"}
    for key, typ in [("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str),
                     ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict), ("paths", list)]:
        assert isinstance(d[key], typ), key
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1
    assert d["scaling"] == "strong" and d["higher_is_better"] is True and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["achieved"] > 0
    assert "traffic" in r and "traffic_profiled" in r and "traffic_source" in r
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1 and c["unit"] == d["unit"] and c["sample"]
    assert c["value_f32"] > 0 and c["arithmetic"] == "f64"
    assert d["recall_at_1"] == 1.0 and d["topk_index_agreement_vs_oracle"] == 1.0
    assert abs(d["value"] - 256 * 4 / (d["ms_per_step"] * 4 / 1e3)) / d["value"] < 1e-6
    # one rank's step of a 2 / 4 / 8-GPU run, emulated on this GPU with the other ranks' real parts: the merged result is the
    # one-GPU result
    emu = d["multi_gpu_emulation"]
    assert [e["ranks"] for e in emu] == [2, 4, 8]
    assert all(e["merged_result_equals_one_gpu"] is True and e["ms_per_batch_per_rank"] > 0 and "EMULATED" in e["label"] for e in emu)
    # the other rows of the hot path, each with its own roofline and CPU baseline, each checked against the oracle
    paths = {p["path"]: p for p in d["paths"]}
    assert set(paths) == {"SDAV.transform", "SDAV.transform (f16x2 split, tolerance mode)",
                          "SDAV 4096-wide variant (non-reference): encode + cosine top-20 of all patch descriptors",
                          "SDAV.train_step (layer 0, 10 frames)", "SDAV similarity matrix",
                          "SDAV similarity matrix, real-frame statistics, N(0,1) weights",
                          "SDAV similarity matrix, real-frame statistics, 1/sqrt(fan_in) weights",
                          "patch front-end (grey + Harris + 30 patches of 41x41)",
                          "LoopClosureDetector.query_and_insert (batches of 32 frames)",
                          "SdavLoopClosureDetector.query_and_insert (batches of 32 frames)",
                          "SdavLoopClosureDetector.submit / result (batches of 32 frames, two in flight)",
                          "cosine similarity matrix (flattened SDAV descriptors)",
                          "cosine top-20 (flattened SDAV descriptors)", "CnnVtl.transform", "cnn_vtl distance matrix",
                          "configs[1] end to end: 24 frames -> patches -> SDAV -> similarity matrix, fp64 encoder (parity mode)",
                          "configs[1] end to end: 24 frames -> patches -> SDAV -> similarity matrix, f16x2 encoder (tolerance mode)",
                          "configs[2] end to end: 24 frames -> CnnVtl -> distance matrix",
                          "configs[0] end to end: 20 real frames -> patches -> SDAV -> 20 x 20 cosine matrix"}
    # BASELINE configs[1] / configs[2] as one device-resident path each: the matrix of the composed call is the staged calls'
    for name in [n_ for n_ in paths if n_.startswith(("configs[1]", "configs[2]"))]:
        row = paths[name]
        assert row["equals_staged_calls_bit_for_bit"] is True and row["device_resident_ms"] > 0 and row["host_to_host_ms"] > 0
        assert abs(sum(row["stage_ms_device_resident"].values()) - row["device_resident_ms"]) < 0.5 * row["device_resident_ms"] + 1.0
    assert paths["configs[2] end to end: 24 frames -> CnnVtl -> distance matrix"]["bit_exact_vs_oracle_on_the_sample"] is True
    assert paths["configs[1] end to end: 24 frames -> patches -> SDAV -> similarity matrix, fp64 encoder (parity mode)"][
        "matrix_agreement_with_oracle_on_the_sample"] > 0.99
    assert paths["configs[0] end to end: 20 real frames -> patches -> SDAV -> 20 x 20 cosine matrix"]["max_abs_err_vs_oracle"] < 2e-3
    # BASELINE configs[3] / configs[4]: a timed one-GPU row each + rank 0's step of the 8-GPU form, merged result == one-shot result
    cfg = {c_["config"]: c_ for c_ in d["baseline_configs"]}
    assert set(cfg) == {3, 4} and cfg[3]["dtype"] == "bf16" and cfg[4]["dtype"] == "f16" and cfg[4]["db_rows"] == 30000
    for c_ in cfg.values():
        assert c_["value"] > 0 and c_["ms_per_step"] > 0 and c_["recall_at_1_of_planted_rows_in_this_db"] == 1.0
        assert 0 < c_["roofline"]["frac"] < 1 and c_["roofline"]["kernel_ms"] <= c_["ms_per_step"] * 1.001
        e8 = c_["eight_gpu_emulation"]
        assert e8["ranks"] == 8 and e8["merged_result_equals_one_gpu"] is True and "EMULATED" in e8["label"]
    assert d["roofline"]["frac_of_measured_copy"] > d["roofline"]["frac"]
    assert paths["SDAV.transform (f16x2 split, tolerance mode)"]["rel_l2_vs_fp64_encoder_max"] < 1e-4
    for name in ("N(0,1) weights", "1/sqrt(fan_in) weights"):
        row = paths["SDAV similarity matrix, real-frame statistics, " + name]
        assert row["equals_fp64_route_bit_for_bit"] is True and row["filter_took_the_call"] is True and row["stats"][1] == 0
    assert paths["SDAV.train_step (layer 0, 10 frames)"]["loss_rel_err_vs_oracle"] < 1e-9
    assert paths["patch front-end (grey + Harris + 30 patches of 41x41)"]["bit_exact_vs_oracle"] is True
    assert paths["LoopClosureDetector.query_and_insert (batches of 32 frames)"]["index_agreement_vs_oracle"] > 0.999
    assert paths["SdavLoopClosureDetector.query_and_insert (batches of 32 frames)"]["stream_poisoned"] == 0
    assert paths["SdavLoopClosureDetector.submit / result (batches of 32 frames, two in flight)"]["same_lists_as_batch_by_batch"] is True
    for p in d["paths"]:
        pr, pc = p["roofline"], p["cpu_baseline"]
        assert p["frames"] == (10 if "train_step" in p["path"] else 20 if p["path"].startswith("configs[0]") else 24)
        assert p["value"] > 0 and p["ms"] > 0 and p["reference"]
        assert pr["bound"] in ("hbm", "mfma") and pr["achieved"] > 0 and abs(pr["frac"] - pr["achieved"] / pr["peak"]) < 1e-9
        assert pr["kernel_ms"] > 0 and pr["kernel_ms"] <= pr["call_ms"] * 1.001 and "traffic" in pr
        assert pc["kind"] == "port" and pc["value"] > 0 and pc["cores"] >= 1 and pc["unit"] == p["unit"] and pc["sample"]
    assert paths["SDAV.transform"]["max_abs_err_vs_oracle"] < 1e-9
    wide = paths["SDAV 4096-wide variant (non-reference): encode + cosine top-20 of all patch descriptors"]
    assert wide["max_abs_err_vs_oracle"] < 1e-9 and wide["topk_index_agreement_vs_oracle"] == 1.0 and wide["dim"] == 4096
    assert paths["SDAV similarity matrix"]["max_rel_err_vs_oracle"] < 1e-9
    assert paths["cosine similarity matrix (flattened SDAV descriptors)"]["max_abs_err_vs_oracle"] < 2e-5
    top = paths["cosine top-20 (flattened SDAV descriptors)"]
    assert top["topk_index_agreement_vs_oracle"] == 1.0 and top["topk_score_max_abs_err_vs_oracle"] < 1e-12
    assert top["queries_resolved_by_exhaustive_pass"] >= 0
    assert paths["CnnVtl.transform"]["int8_bytes_differing_from_oracle"] == 0
    assert paths["cnn_vtl distance matrix"]["bit_exact_vs_oracle"] is True
    assert paths["SDAV.transform"]["roofline"]["kernel_launches_timed"] == 5


def test_bench_two_ranks_equal_one_rank():
    """`bench.py --gpus 2` as the driver launches it (python -m torch.distributed.run, one process per rank),
    with --backend gloo --share-gpu so that both ranks can use this box's single GPU: MatchPipeline's sharded
    protocol (group selection -> all-gather of the group maxima -> filtered re-score -> all-gather of the
    packed per-shard top-k -> merge) must give recall 1.0 and exactly the one-rank result."""
    common = ["--steps", "6", "--warmup", "2", "--rows", "40000", "--no-cpu-baseline", "--no-power-probe", "--no-paths", "--no-rccl-smoke"]
    one = run_bench(["--gpus", "1"] + common)
    port = 29600 + os.getpid() % 300
    two = run_bench(["--gpus", "2", "--backend", "gloo", "--share-gpu"] + common,
                    launcher=[sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                              "--master-addr", "127.0.0.1", "--master-port", str(port)])
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["config"]["pipelined"] is True
    assert two["config"]["rows_per_gpu"] == 20000 and one["config"]["rows_per_gpu"] == 40000
    assert one["recall_at_1"] == 1.0 and two["recall_at_1"] == 1.0
    assert two["topk_idx_sha256"] == one["topk_idx_sha256"]
    assert two["topk_scores_sha256"] == one["topk_scores_sha256"]
    assert two["steps"] == 6 and two["value"] > 0 and two["scaling"] == "strong"
    # the collectives were exercised and the pipeline checked against a plain exchange before anything was timed
    sm = two["rccl_smoke"]
    assert two["rccl_ranks"] == 2 and sm["ranks"] == 2 and sm["pipeline_equals_plain_exchange"] is True
    assert sm["collective_us"]["batches"] == 3 and sm["collective_us"]["group_maxima"] > 0
    assert one["rccl_smoke"] is None and one["collective_us"] is None


def test_bench_launches_its_own_ranks():
    """A bare `python bench.py --gpus 2` (no torch.distributed.run around it -- how a driver that only knows the N = 1
    command line would call it) starts its two rank processes itself, before it touches the GPU, relays rank 0's one
    JSON line and exit code, and gets the one-rank digests.  Four ranks on the one GPU as well (12 500-row shards: the
    small-database plan on every rank, another plan than the one-rank run's -- same digests)."""
    common = ["--steps", "4", "--warmup", "1", "--rows", "50000", "--no-cpu-baseline", "--no-power-probe", "--no-paths", "--no-rccl-smoke"]
    one = run_bench(["--gpus", "1"] + common)
    for ranks in (2, 4):
        got = run_bench(["--gpus", str(ranks), "--backend", "gloo", "--share-gpu"] + common)
        assert got["n_gpus"] == ranks and got["rccl_smoke"]["pipeline_equals_plain_exchange"] is True
        assert got["recall_at_1"] == 1.0
        assert got["topk_idx_sha256"] == one["topk_idx_sha256"] and got["topk_scores_sha256"] == one["topk_scores_sha256"]


def test_sharded_exhaustive_round_across_processes():
    """`bench.py --crowded`: query 0's planted row copied to kg * 8 + 1 places spread over every shard -- its k-th score
    ties with a row every selection leaves behind, so NO merge can certify it.  One rank resolves it inside its one-shot
    call; two real processes (gloo, sharing the GPU) must take MatchPipeline._resolve -- the exhaustive pass on every rank,
    a third all-gather, a merge -- for every batch, agree on the flags (nobody hangs in a collective the other skipped),
    drop nothing and return exactly the one-rank lists."""
    common = ["--steps", "5", "--warmup", "2", "--rows", "40000", "--no-cpu-baseline", "--no-power-probe", "--no-paths",
              "--no-shard-emulation", "--no-rccl-smoke", "--crowded"]
    one = run_bench(["--gpus", "1"] + common)
    port = 29900 + os.getpid() % 90
    two = run_bench(["--gpus", "2", "--backend", "gloo", "--share-gpu"] + common,
                    launcher=[sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                              "--master-addr", "127.0.0.1", "--master-port", str(port)])
    assert one["config"]["crowded"] is True and two["config"]["crowded"] is True and two["n_gpus"] == 2
    p = two["pipeline"]
    assert p["resolved_batches"] >= 5 and p["dropped_batches"] == 0            # every timed batch went through the round
    assert two["rccl_smoke"]["pipeline_equals_plain_exchange"] is True and two["rccl_smoke"]["resolved_batches"] == 3
    assert two["topk_idx_sha256"] == one["topk_idx_sha256"] and two["topk_scores_sha256"] == one["topk_scores_sha256"]
    assert one["recall_at_1"] == 1.0 and two["recall_at_1"] == 1.0


def test_rccl_first_contact_on_one_gpu():
    """`python -m deeploopcloser_amd.dist --world1-smoke` in a fresh process: a one-rank "nccl" (= RCCL) process group, the
    sharded protocol with every collective forced through the library on the second stream (MatchPipeline(...,
    force_collectives=True): norm all-reduce, both all-gathers, the certifying merge; and, on a crowded database, the
    exhaustive round with its third all-gather) -- every batch equal to the one-shot call bit for bit.  What a first
    `--gpus 8` run brings up, minus the other seven ranks."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    res = subprocess.run([sys.executable, "-m", "deeploopcloser_amd.dist", "--world1-smoke", "--rows", "65536"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-3000:])
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert d["ok"] is True and d["backend"] == "nccl" and d["world"] == 1
    assert d["pipeline_equals_one_shot"] is True and d["crowded_equals_one_shot"] is True
    assert d["pipeline_resolved_batches"] == 0 and d["crowded_resolved_batches"] == 6
    c = d["pipeline_collective_us"]
    assert c["batches"] == 6 and c["group_maxima"] > 0 and c["packed_topk"] > 0
