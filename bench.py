#!/usr/bin/env python3
"""Headline benchmark: query-frames/sec against an N-keyframe descriptor DB.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...        (no launcher: starts the N rank processes itself, before touching the GPU)

Workload (BASELINE.json north_star / configs[4]): a 1,000,000-row synthetic
4096-d key-frame database stored as L2-normalised bf16, batches of 256 query
descriptors, top-20 cosine match.  The database is row-sharded over the N GPUs
(total size fixed -> "strong" scaling); one step = one query batch scored
against the WHOLE database: score pass on each shard, RCCL all-gather of the
shards' group maxima, fp64 re-score of the groups that can hold a global top-k
row, RCCL all-gather of the per-shard [256,20] parts, certifying fp64 merge
(deeploopcloser_amd.matching.MatchPipeline).  Database and queries are resident
in HBM before the timed region.  With N > 1 the collectives are exercised and
the pipeline's result is checked against a second, independent exchange BEFORE
anything is timed (`rccl_smoke` in the line).

Rank 0 prints ONE JSON line (the driver's contract) carrying also
  roofline     -- the dominant kernel (the MFMA score GEMM): algorithmic bytes
                  per launch / mean launch duration from HIP events recorded on
                  the launch stream inside the timed region;
  cpu_baseline -- the CPU oracle (oracle/cosine.py, a port: fp64 NumPy) timed
                  on this host's cores on a bounded sample at N=1;
  paths        -- (N=1 only, outside the timed region) the other rows of the hot path at
                  BASELINE configs[1] / configs[2] size (1063 frames): SDAV.transform, the SDAV
                  similarity matrix, the cosine matrix / top-20 over the flattened SDAV
                  descriptors, CnnVtl.transform, the cnn_vtl distance matrix -- each with its own
                  roofline (dominant-kernel time from HIP events on the launch stream) and
                  cpu_baseline (the oracle on a bounded sample of the same input).
"""
import argparse
import json
import os
import re
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # this pool's driver only has dmabuf IPC (RCCL needs it)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
HBM_COPY_GBS = 6290.0        # MI355X_MICROARCH.md / SURVEY 8d: what a device copy measures on this part
MFMA_PEAK_TFLOPS = 2500.0    # dense bf16 / fp16 MFMA peak
MFMA_F64_PEAK_TFLOPS = 78.6  # dense fp64 MFMA peak (v_mfma_f64_16x16x4_f64)
MFMA_I8_PEAK_TOPS = 5000.0   # dense int8 MFMA peak (v_mfma_i32_16x16x64_i8: twice the bf16 form's work per clock)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows", type=int, default=1_000_000, help="key-frames in the WHOLE database")
    ap.add_argument("--dim", type=int, default=4096)
    ap.add_argument("--queries", type=int, default=256)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-power-probe", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --share-gpu rehearses the N>1 path with several ranks on ONE GPU")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks use cuda:0 (rehearsal only)")
    ap.add_argument("--pipeline", action="store_true",
                    help="two-stream MatchPipeline also on one GPU (default there: one-shot calls, measured 3 %% faster "
                         "-- at the power cap the overlapped selection costs the GEMM more than it hides)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="one stream, no overlap of a batch's selection / all-gather with the next batch's GEMM")
    ap.add_argument("--cpu-sample-rows", type=int, default=1_000_000)
    ap.add_argument("--no-paths", action="store_true", help="skip the `paths` entries (configs[1] / configs[2] rows)")
    ap.add_argument("--no-shard-emulation", action="store_true",
                    help="skip `multi_gpu_emulation` (one rank's step of a 2 / 4 / 8-GPU run, emulated on this GPU)")
    ap.add_argument("--path-frames", type=int, default=1063, help="frames of the `paths` entries (outdoor_kennedylong: 1063)")
    ap.add_argument("--no-configs", action="store_true", help="skip the `baseline_configs` rows (configs[3], configs[4])")
    ap.add_argument("--no-rccl-smoke", action="store_true",
                    help="N=1: skip the child process that brings RCCL up on this GPU (one-rank nccl group, the sharded "
                         "protocol with its collectives forced through the library) after everything else has been measured")
    ap.add_argument("--detail", default=None, help="where the full result goes (default: bench_detail.json beside this "
                    "script, and gpurun_out/ when present); stdout carries one line under 4 KB")
    ap.add_argument("--crowded", action="store_true",
                    help="rehearsal of the sharded exhaustive round: the database row query 0 is planted on is copied to kg*8+1 "
                         "places spread over the whole database (every shard), so that query's k-th score ties with a row every "
                         "selection leaves behind -- no merge can certify it and every batch goes through MatchPipeline._resolve")
    return ap.parse_args()


def blas_threads():
    """Threads of the BLAS pool NumPy's matmul runs on (what the fp64 oracles are timed with)."""
    try:
        from threadpoolctl import threadpool_info
        n = [p.get("num_threads", 0) for p in threadpool_info() if p.get("user_api") == "blas"]
        if n:
            return int(max(n))
    except Exception:
        pass
    return int(os.cpu_count() or 1)


def pmc_traffic(n, d, nq, dtype, world):
    """HBM bytes per launch of the score GEMM from the latest committed rocprofv3 PMC summary
    (profiles/*_pmc_summary.json, collected with scripts/collect_profiles.sh in separate --pmc
    passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950), if one exists for
    exactly this workload; bench.py itself cannot run the profiler."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))):
        try:
            g = json.load(open(f))["kernels"]["score_gemm_kernel"]
        except Exception:
            continue
        w = g.get("workload", {})
        if (w.get("db_rows"), w.get("dim"), w.get("queries"), w.get("dtype"), w.get("n_gpus")) == (n, d, nq, dtype, world):
            best = (g.get("hbm_traffic_bytes_per_launch"), os.path.basename(f))
    return best


def _smi(*flags):
    import subprocess
    return subprocess.run(["rocm-smi"] + list(flags), capture_output=True, text=True, timeout=10).stdout


def _energy_uj():
    """The package's accumulated-energy counter in microjoules (rocm-smi --showenergycounter), or None."""
    try:
        m = re.search(r"Accumulated Energy \(uJ\):\s*([0-9.]+)", _smi("--showenergycounter"))
        return float(m.group(1)) if m else None
    except Exception:
        return None


def power_probe(step, seconds=2.5):
    """Keep submitting steps for `seconds` while a thread samples `rocm-smi --showpower --showclocks`
    (a child process); returns the median package power (W) and shader clock (MHz), the steps run and their
    mean duration, and -- when the board exposes its energy accumulator -- the counter's joules per step over
    the same stretch (the two reads sit outside the loop, behind a synchronise).  None if rocm-smi is missing."""
    import threading
    samples, stop = [], threading.Event()

    def sample():
        while not stop.is_set():
            try:
                txt = _smi("--showpower", "--showclocks")
                pw = re.search(r"Power \(W\):\s*([0-9.]+)", txt)
                sc = re.search(r"sclk clock level:.*?\((\d+)Mhz\)", txt)
                if pw and sc:
                    samples.append((float(pw.group(1)), int(sc.group(1))))
            except Exception:
                return
            stop.wait(0.3)

    try:
        for _ in range(50):                                  # the chip is under the load before the first counter read
            step()
        torch.cuda.synchronize()
        e0 = _energy_uj()
        th = threading.Thread(target=sample, daemon=True)
        th.start()
        steps = 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for _ in range(50):
                step()
            steps += 50
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        e1 = _energy_uj()
        stop.set()
        th.join(timeout=15)
    except Exception:
        return None
    if len(samples) < 2:
        return None
    good = samples[1:]                                   # the first sample may predate the load
    out = {"package_power_w": float(np.median([g[0] for g in good])), "sclk_mhz": float(np.median([g[1] for g in good])),
           "samples": len(good), "steps": steps, "ms_per_step": dt / steps * 1e3,
           "source": "rocm-smi while the timed loop's step keeps running (untimed)"}
    if e0 is not None and e1 is not None and e1 > e0:
        out["energy_counter_j_per_step"] = (e1 - e0) * 1e-6 / steps
    return out


def synth_shard(eng, n_total, dim, lo, hi, dtype, planted_rows, chunk=32768):
    """Synthetic DB: rows ~ U(0,1)^D, mean-centred, L2-normalised, stored in
    `dtype`.  Every rank walks ALL chunks with the same per-chunk seeds so that it can keep
    its own rows [lo,hi) and also pick up the fp32 rows the queries are planted on."""
    rows = torch.empty((hi - lo, dim), dtype=dtype, device=eng.device)
    planted = torch.empty((len(planted_rows), dim), dtype=torch.float32, device=eng.device)
    prow = torch.as_tensor(planted_rows, device=eng.device)
    for c0 in range(0, n_total, chunk):
        c1 = min(c0 + chunk, n_total)
        need_rows = c1 > lo and c0 < hi
        sel = torch.nonzero((prow >= c0) & (prow < c1)).flatten()
        if not need_rows and sel.numel() == 0:
            continue
        g = torch.Generator(device=eng.device)
        g.manual_seed(1234 + c0 // chunk)
        x = torch.rand((c1 - c0, dim), generator=g, device=eng.device, dtype=torch.float32)
        if sel.numel():
            planted[sel] = x[prow[sel] - c0]
        if need_rows:
            a, b = max(c0, lo), min(c1, hi)
            rows[a - lo:b - lo] = eng.normalize(x[a - c0:b - c0], dtype, center=True)
        del x
    return rows, planted


def _timed_path(eng, fn, reps=3, inner=3):
    """fn() on the current stream: (wall ms of a call from stream events, summed ms of its dense-GEMM launches from the
    HIP events the library records around each of them, number of launches, result) -- per call, as the MEAN over `inner`
    calls back to back, of the fastest of `reps` such runs after one warm-up.  (One call between two synchronisations
    measured the clock the chip idles down to between them as much as the kernels: the tolerance-mode encoder read 4.5
    ms a call that way and 4.25 in any loop, docs/LAB.md 11.5b.)"""
    fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(reps):
        eng.set_profiling(True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            r = fn()
        e1.record()
        torch.cuda.synchronize()
        k = eng.profile_gemm_ms(256)
        eng.set_profiling(False)
        whole = len(k) % inner == 0 and len(k) < 256              # (a ring that wrapped cannot be split by call)
        cur = (e0.elapsed_time(e1) / inner, (float(np.sum(k)) / inner if whole else None) if k else None,
               len(k) // inner if whole else 0, r)
        if best is None or cur[0] < best[0]:
            best = cur
    return best


def _timed_host(fn, reps=3):
    """Wall-clock ms (host to host: the call returns NumPy data) of the fastest of `reps` calls after one warm-up."""
    fn()
    best, r = None, None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        dt = (time.perf_counter() - t0) * 1e3
        best = dt if best is None or dt < best else best
    return best, r


def _mfma_f64_roofline(flops, kernel_ms, launches, call_ms, kernel):
    tf = flops / (kernel_ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": tf, "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": tf / MFMA_F64_PEAK_TFLOPS, "traffic": None, "kernel": kernel, "kernel_ms": kernel_ms,
            "kernel_launches_timed": launches, "call_ms": call_ms, "algorithmic_flops_per_call": flops}


def bench_paths(eng, n_frames):
    """The rows of the hot path other than the headline match, at BASELINE configs[1] / configs[2] size, each
    with the roofline of its dominant kernel and the CPU oracle timed on a bounded sample of the same input
    (the reference's Python cannot travel to this box; oracle/ restates it, `kind: port`)."""
    import gc
    import deeploopcloser_amd as dlc
    from oracle import sdav as osdav, similarity as osim, distance as odist, cnn_vtl as ocnn, cosine as ocos
    gc.collect()                                                         # (what the headline part left behind goes now, not inside a timed row)
    torch.cuda.empty_cache()                                             # (... and the allocator's cached 8 GB blocks of the config rows)
    N, P, K0, H = n_frames, 30, 1681, 2500
    cores = blas_threads()
    out = []
    g = torch.Generator(device=eng.device)
    g.manual_seed(0)

    # ---- E1/E2: SDAV.transform (SDAV.py:293-302), fp64 like the reference ----------------------------------
    x = torch.rand((N, P, K0), generator=g, device=eng.device, dtype=torch.float64)
    net = dlc.SDAV(seed=1)
    call_ms, k_ms, k_n, h = _timed_path(eng, lambda: net.transform_tensor(x))
    flops = 2.0 * P * N * (K0 * H + 4 * H * H)
    ws, bs = net.get_weights()
    nb = min(N, 64)
    xs = x[:nb].cpu().numpy()
    t0 = time.perf_counter()
    ref = osdav.transform(xs, ws, bs)
    t_cpu = time.perf_counter() - t0
    err = float(np.abs(h[:nb * P].cpu().numpy() - ref).max())
    # the reference's own contract: ndarray in, ndarray out (pageable host arrays both ways: 429 MB in, 638 MB out)
    x_np = x.cpu().numpy()
    h2h_ms, h_np = _timed_host(lambda: net.transform(x_np))
    h2h_same = bool(np.array_equal(h_np, h.cpu().numpy()))
    del x_np, h_np
    out.append({"path": "SDAV.transform", "reference": "src/sdav/network/SDAV.py:293-302", "frames": N, "dtype": "f64",
                "value": N / (call_ms * 1e-3), "unit": "frames/s", "ms": call_ms,
                "host_to_host_ms": h2h_ms, "host_to_host_bit_identical": h2h_same,
                "roofline": _mfma_f64_roofline(flops, k_ms, k_n, call_ms, "gemm_dma_f64_kernel (5 layers, LDS-DMA fp64 GEMM, fused bias + sigmoid)"),
                "cpu_baseline": {"value": nb / t_cpu, "unit": "frames/s", "cores": cores, "kind": "port",
                                 "sample": "oracle/sdav.py (fp64 NumPy) on the first %d of the %d frames, same weights: "
                                           "%.1f s of CPU work" % (nb, N, t_cpu)},
                "max_abs_err_vs_oracle": err})
    # ---- E1/E2 in the TOLERANCE mode (opt-in): three fp16 MFMA products of two-piece splits per layer, fp32 accumulate
    # (csrc/gemm_split_f16.hip); same weights, same frames; error against the fp64 encoder above
    net16 = dlc.SDAV(seed=1, dtype="f16x2")
    net16.transform_tensor(x[:2])                                        # (prepares the weight panels once)
    s_ms, sk_ms, sk_n, h16 = _timed_path(eng, lambda: net16.transform_tensor(x))
    l2 = (h16 - h).norm(dim=1) / h.norm(dim=1)
    sflops = 3.0 * flops                                                 # three 16-bit products per fp64-equivalent product
    out.append({"path": "SDAV.transform (f16x2 split, tolerance mode)", "reference": "src/sdav/network/SDAV.py:126-163,293-302",
                "frames": N, "dtype": "f16x2 (two fp16 pieces per operand, fp32 accumulate, fp64 in / out)",
                "value": N / (s_ms * 1e-3), "unit": "frames/s", "ms": s_ms,
                "rel_l2_vs_fp64_encoder_max": float(l2.max()), "rel_l2_vs_fp64_encoder_median": float(l2.median()),
                "tolerance": "north_star: descriptor L2 within 1e-4", "speedup_vs_fp64_mode": call_ms / s_ms,
                "roofline": {"bound": "mfma", "achieved": sflops / (sk_ms * 1e-3) / 1e12, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": sflops / (sk_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, "traffic": None,
                             "kernel": "gemm_split_f16_kernel (5 layers x 3 fp16 products, LDS-DMA rings, fused bias + sigmoid + re-split)",
                             "kernel_ms": sk_ms, "kernel_launches_timed": sk_n, "call_ms": s_ms,
                             "algorithmic_flops_per_call": sflops, "fp64_equivalent_tflops": flops / (sk_ms * 1e-3) / 1e12},
                "cpu_baseline": out[-1]["cpu_baseline"]})
    del net16, h16, l2
    del ws, bs, ref

    # ---- SURVEY 8d config 2's optional variant, NOT the reference's network: hidden_units[-1] = 4096 -- north_star's
    # "4096-d descriptors" at configs[1] size: encode (fp64, the same chain, last layer 2500 -> 4096), then every one of the
    # N * 30 patch descriptors matched against all of them (cosine top-20, bf16, batches of 256 queries)
    from oracle import cosine as ocos4
    W4 = 4096
    net4 = dlc.SDAV(seed=1, hidden_units=[H, H, H, H, W4])
    e_ms, ek_ms, ek_n, h4 = _timed_path(eng, lambda: net4.transform_tensor(x), reps=2)
    eflops = 2.0 * P * N * (K0 * H + 3 * H * H + H * W4)
    db4 = dlc.KeyframeDatabase(h4, dtype="bf16", center=True)
    q4 = db4.rows
    nq4 = q4.shape[0]

    QB4 = 4096                                                           # query rows per call: the score pass walks the database
                                                                         # once per 256 of them, sixteen query blocks sharing
    def all_patches_top20():                                             # every database tile out of one XCD's L2
        r = None
        for b0 in range(0, nq4, QB4):
            r = db4.match_topk(q4[b0:b0 + QB4], 20)
        return r
    m_ms, _, _, _ = _timed_path(eng, all_patches_top20, reps=2, inner=1)
    ws4, bs4 = net4.get_weights()
    nb4 = min(N, 16)
    t0 = time.perf_counter()
    ref4 = osdav.transform(xs[:nb4], ws4, bs4)
    t_cpu4 = time.perf_counter() - t0
    err4 = float(np.abs(h4[:nb4 * P].cpu().numpy() - ref4).max())
    nqs = min(64, nq4)
    rows4_h = q4.float().cpu().numpy().astype(np.float64)
    t0 = time.perf_counter()
    es4, ei4 = ocos4.cosine_topk(rows4_h[:nqs], rows4_h, 20)
    t_cos4 = time.perf_counter() - t0
    top4 = db4.match_topk(q4[:nqs], 20)
    tot4 = e_ms + m_ms
    r4 = _mfma_f64_roofline(eflops, ek_ms, ek_n, tot4, "dominant stage: the encoder's five fp64 GEMMs (gemm_dma_f64_kernel); "
                            "the match is score_gemm_kernel + finish_topk_kernel per call of 4096 patch queries")
    out.append({"path": "SDAV 4096-wide variant (non-reference): encode + cosine top-20 of all patch descriptors",
                "reference": "NOT the reference's network (SDAV.py:31-32 fixes 5 x 2500): SURVEY 8d config 2's optional "
                "hidden_units[-1] = 4096 variant, north_star's 4096-d descriptors", "frames": N, "dtype": "f64 encode, bf16 cosine",
                "dim": W4, "k": 20, "value": N / (tot4 * 1e-3), "unit": "frames/s", "ms": tot4,
                "stage_ms": {"SDAV.transform (last layer 4096 wide)": e_ms, "cosine top-20, %d x %d patch descriptors, %d queries per call"
                             % (nq4, nq4, QB4): m_ms},
                "patch_queries_per_s": nq4 / (m_ms * 1e-3), "roofline": r4,
                "cpu_baseline": {"value": 1.0 / (t_cpu4 / nb4 + (t_cos4 / nqs) * P), "unit": "frames/s", "cores": cores, "kind": "port",
                                 "sample": "oracle/sdav.py on the first %d frames (%.2f s) + oracle/cosine.py top-20 of the first %d "
                                           "patch descriptors against all %d (%.2f s), per frame" % (nb4, t_cpu4, nqs, nq4, t_cos4)},
                "max_abs_err_vs_oracle": err4,
                "topk_index_agreement_vs_oracle": float((top4[1].cpu().numpy() == ei4).mean())})
    del net4, h4, db4, q4, rows4_h, ws4, bs4, ref4, top4, xs

    # ---- f-2: one SDAV training step (sess.run(train_steps[0]), SDAV.py:262) on the reference's default batch -------
    from oracle import sdav_train as otrain
    B = min(10, N)                                                       # SDAV.default_batch_size
    if B >= 2:
        tnet = dlc.SDAV(seed=3)
        xb = x[:B].contiguous()
        masks = [tnet._mask(0)]
        w0, be0 = [w.cpu().numpy() for w in tnet._weights], [b.cpu().numpy() for b in tnet._biases]
        bd0 = tnet._biases_dec[0].cpu().numpy()
        t0 = time.perf_counter()
        want = otrain.loss_and_grads(0, xb.cpu().numpy(), [masks[0].cpu().numpy()], w0, be0, bd0, tnet.sparse_level,
                                     tnet.sparse_penalty, tnet.consecutive_penalty)[0]
        t_cpu = time.perf_counter() - t0
        with eng.latency_mode():                                          # as SDAV.fit / fit_dataset run their steps: split-K scratch on
            got = float(tnet.train_step(0, xb, masks)[0].item())          # the loss of the first step: same parameters as the oracle's
            # the steps as SDAV.fit / fit_dataset run them: one captured HIP graph replayed per step, masks redrawn in place
            tnet.train_steps(0, xb, 5)
            torch.cuda.synchronize()
            # the fastest of five runs of 10 steps: a step is 0.5 ms, and one host pause inside a single long run -- Python's
            # cyclic collector freeing what earlier rows left behind took 76 ms once -- would be booked as step time
            steps_t = 10
            step_ms = None
            for _ in range(5):
                t0 = time.perf_counter()
                tnet.train_steps(0, xb, steps_t)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / steps_t * 1e3
                step_ms = dt if step_ms is None or dt < step_ms else step_ms
            t0 = time.perf_counter()
            for _ in range(20):
                tnet.train_step(0, xb, masks)                             # ... and one eager call per step (19 launches each)
            torch.cuda.synchronize()
            eager_ms = (time.perf_counter() - t0) / 20 * 1e3
        tflops = 5 * 2.0 * (B * P) * K0 * H                               # encoder, decoder, dh, and the two weight gradients
        tf = tflops / (step_ms * 1e-3) / 1e12
        out.append({"path": "SDAV.train_step (layer 0, %d frames)" % B, "reference": "src/sdav/network/SDAV.py:129-226, 262",
                    "frames": B, "dtype": "f64", "value": 1e3 / step_ms, "unit": "steps/s", "ms": step_ms,
                    "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                 "frac": tf / MFMA_F64_PEAK_TFLOPS, "traffic": None,
                                 "kernel": "five fp64 GEMMs of 300 x 1681 x 2500 (split-K, as SDAV.fit runs them) + 7 small kernels per step (r04: 14; "
                                           "the column sums, every SGD update and the loss are one launch), captured once as a HIP graph "
                                           "and replayed (SDAV.train_steps), masks redrawn in place per step",
                                 "kernel_ms": step_ms, "call_ms": step_ms, "algorithmic_flops_per_call": tflops,
                                 "eager_ms_per_step": eager_ms},
                    "cpu_baseline": {"value": 1.0 / t_cpu, "unit": "steps/s", "cores": cores, "kind": "port",
                                     "sample": "oracle/sdav_train.py loss_and_grads (fp64 NumPy) on the same batch, masks and "
                                               "parameters: %.2f s" % t_cpu},
                    "loss_rel_err_vs_oracle": abs(got - want) / abs(want)})
        del tnet, xb, masks

    # ---- f-1: the patch front-end on resident uint8 frames (CvInputParser.py:19-49 with the build's detector) -------
    from oracle import patches as opatch, keypoints as okp
    from deeploopcloser_amd.input import CvInputParser
    fh, fw = 192, 240
    rgb = torch.randint(0, 256, (N, fh, fw, 3), generator=g, device=eng.device, dtype=torch.uint8)
    parser = CvInputParser(P, 41)
    fe_ms, _, _, pat = _timed_path(eng, lambda: parser.parse_batch(rgb))
    fe_bytes = float(N) * (fh * fw * 3 + fh * fw + P * K0 * 8)           # RGB in, grey out and in again, fp64 patches out
    nf = min(N, 3)
    pat_h = pat[:nf].cpu().numpy()
    t0 = time.perf_counter()
    fe_exact = True
    for f in range(nf):
        gray = opatch.bgr2gray_opencv(rgb[f].cpu().numpy())
        pts, _, cnt = okp.key_points(gray, P)
        if cnt == P:                                                     # (a frame with fewer corners is topped up with grid points)
            fe_exact = fe_exact and bool(np.array_equal(pat_h[f], opatch.parse(gray, [tuple(q) for q in pts], 41)))
    t_cpu = time.perf_counter() - t0
    out.append({"path": "patch front-end (grey + Harris + %d patches of 41x41)" % P,
                "reference": "src/sdav/input/CvInputParser.py:19-49 (SURF replaced by the build's integer Harris detector)",
                "frames": N, "dtype": "u8", "value": N / (fe_ms * 1e-3), "unit": "frames/s", "ms": fe_ms,
                "roofline": {"bound": "hbm", "achieved": fe_bytes / (fe_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": fe_bytes / (fe_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                             "kernel": "rgb_to_gray + harris (response, non-maximum suppression, n arg-max rounds) + patch gather",
                             "kernel_ms": fe_ms, "call_ms": fe_ms, "algorithmic_bytes_per_call": fe_bytes},
                "cpu_baseline": {"value": nf / t_cpu, "unit": "frames/s", "cores": 1, "kind": "port",
                                 "sample": "oracle/patches.py grey conversion + oracle/keypoints.py Harris on the first %d frames "
                                           "(NumPy): %.2f s" % (nf, t_cpu)},
                "bit_exact_vs_oracle": fe_exact})
    del rgb, pat

    # ---- f-4: the streaming loop-closure query over a growing key-frame database ---------------------------------
    from oracle import loop_closure as oloop
    Dl, kl, excl, bl = 4096, 5, 30, 32
    xs_l = torch.randn((N, Dl), generator=g, device=eng.device, dtype=torch.float32)

    def stream():
        det = dlc.LoopClosureDetector(Dl, k=kl, threshold=0.5, exclusion=excl, capacity=max(64, N))
        outs = [det.query_and_insert(xs_l[lo:lo + bl]) for lo in range(0, N, bl)]
        return det, torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
    lc_ms, _, _, (det, ls, li) = _timed_path(eng, stream)

    def stream_single():                                          # the robot's case: frames arrive one at a time
        det_ = dlc.LoopClosureDetector(Dl, k=kl, threshold=0.5, exclusion=excl, capacity=max(64, N))
        outs_ = [det_.query_and_insert(xs_l[lo:lo + 1]) for lo in range(N)]
        return torch.cat([o[1] for o in outs_])
    lc1_ms, _, _, li1 = _timed_path(eng, stream_single, reps=2)
    nl = min(N, 200)
    rows_l = det.db.rows[:nl].float().cpu().numpy().astype(np.float64)
    t0 = time.perf_counter()
    es, ei = oloop.stream_topk(rows_l, kl, excl)
    t_cpu = time.perf_counter() - t0
    lc_agree = float((li[:nl].cpu().numpy() == ei).mean())
    lc_bytes = float(N) * N / 2 * Dl * 2                                 # every frame reads the older key-frames' bf16 rows once
    out.append({"path": "LoopClosureDetector.query_and_insert (batches of %d frames)" % bl,
                "reference": "SURVEY 8f-4 (no reference implementation); oracle/loop_closure.py", "frames": N, "dim": Dl,
                "dtype": "bf16", "k": kl, "value": N / (lc_ms * 1e-3), "unit": "frames/s", "ms": lc_ms,
                "roofline": {"bound": "hbm", "achieved": lc_bytes / (lc_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": lc_bytes / (lc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                             "kernel": "per batch: l2_normalize into the store + one age-limited top-k match (dlc_cosine_topk_older: the split-K score pass, then small_topk_kernel -- partial sums, row selection, fp64 re-score, certificate) -- %d batches of three launches of 6-15 us each: latency-bound at this database size" % (-(-N // bl)),
                             "kernel_ms": lc_ms, "call_ms": lc_ms, "algorithmic_bytes_per_call": lc_bytes},
                "cpu_baseline": {"value": nl / t_cpu, "unit": "frames/s", "cores": cores, "kind": "port",
                                 "sample": "oracle/loop_closure.py per-frame fp64 loop over the first %d stored rows: %.2f s" % (nl, t_cpu)},
                "index_agreement_vs_oracle": lc_agree,
                "one_frame_at_a_time": {"ms": lc1_ms, "us_per_frame": lc1_ms * 1e3 / N, "frames_per_s": N / (lc1_ms * 1e-3),
                                        "same_lists_as_batched": bool(torch.equal(li1, li))}})
    del xs_l, det, ls, li, li1

    # ---- M1/M2: SDAV similarity matrix (SimilarityCalculator.py:12-49 + create_similarity_matrix.py:29-38) ----
    desc = h.reshape(N, P, H)

    def sim():                                                    # what SimilarityCalculator(dataset).similarity_matrix() runs
        score, rng = eng.distinctive_score(desc, 0.5, 0.2, with_range=True)
        return eng.sdav_similarity_matrix(desc, score, 10.0, -10.0, range=rng)
    call_ms, k_ms, k_n, (mf, mi) = _timed_path(eng, sim, reps=4)

    def sim_direct(d_):                                            # [arg-mins evaluated directly, why the fp64 form ran (0: it did not)]
        st_ = torch.zeros((2,), dtype=torch.int64, device=eng.device)
        eng.sdav_similarity_matrix(d_, eng.distinctive_score(d_, 0.5, 0.2), 10.0, -10.0, stats=st_)
        return [int(v) for v in st_.tolist()]
    # The patch products the similarity needs: row patches of frame i against the patches of every LATER frame j (the
    # upper triangle; gram_i8_kernel launches the tiles that hold such a pair and decides the patch arg-min in its
    # epilogue) -- six int8 products of length H per patch pair (csrc/gram_i8.hip: the 24-bit fixed-point digits' classes
    # 2, 3 and 4).  Padding (K to 2560, whole-frame column units, diagonal tiles) is the kernel's cost, not counted here.
    patch_pairs = (N * (N - 1) / 2.0) * P * P
    i8_ops = 6 * 2.0 * patch_pairs * H
    pairs = N * (N - 1) // 2
    ns = min(N, 20)                                           # the reference-literal per-pair loop at datasets/test size
    dsn = desc[:ns].cpu().numpy()
    t0 = time.perf_counter()
    ref = osim.similarity_matrix_f64(dsn)
    t_cpu = time.perf_counter() - t0
    # literal form: the reference recomputes the dataset mean and the distinctive score for EVERY pair (:13-14)
    t0 = time.perf_counter()
    for _ in range(8):
        osim.similarity_score(dsn, dsn[0], dsn[1])
    t_lit = (time.perf_counter() - t0) / 8
    calc = dlc.SimilarityCalculator(dsn)                       # the drop-in class, called once per pair as the reference's loop does
    calc.similarity_score(dsn[0], dsn[1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for e_ in range(50):
        calc.similarity_score(dsn[e_ % ns], dsn[(e_ + 1) % ns])
    t_drop = (time.perf_counter() - t0) / 50
    del calc
    sub = eng.sdav_similarity_matrix(desc[:ns].contiguous(), eng.distinctive_score(desc[:ns].contiguous(), 0.5, 0.2), 10.0, -10.0)[0]
    fin = np.isfinite(ref)
    err = float(np.abs(sub.cpu().numpy()[fin] - ref[fin]).max() / max(1.0, np.abs(ref[fin]).max()))
    out.append({"path": "SDAV similarity matrix", "reference": "src/sdav/similarity/SimilarityCalculator.py:12-49, "
                "src/sdav/create_similarity_matrix.py:29-38", "frames": N, "dtype": "f64",
                "value": pairs / (call_ms * 1e-3), "unit": "frame-pairs/s", "ms": call_ms,
                "roofline": {"bound": "mfma", "achieved": i8_ops / (k_ms * 1e-3) / 1e12, "peak": MFMA_I8_PEAK_TOPS, "unit": "TOP/s",
                             "frac": i8_ops / (k_ms * 1e-3) / 1e12 / MFMA_I8_PEAK_TOPS, "traffic": None,
                             "kernel": "gram_i8_kernel (exact int8 products of the column-centred descriptors' 24-bit fixed-point digits, "
                                       "wanted tiles only; its epilogue decides the patch arg-min, undecided ones are evaluated directly in fp64)",
                             "kernel_ms": k_ms, "kernel_launches_timed": k_n, "call_ms": call_ms,
                             "algorithmic_ops_per_call": i8_ops,
                             "fp64_equivalent_tflops": 2.0 * patch_pairs * H / (k_ms * 1e-3) / 1e12},
                "cpu_baseline": {"value": (ns * (ns - 1) // 2) / t_cpu, "unit": "frame-pairs/s", "cores": cores, "kind": "port",
                                 "sample": "oracle/similarity.py all-vs-all loop on the first %d frames (%d pairs, mean / "
                                           "distinctive score hoisted): %.2f s; the literal similarity_score (mean recomputed "
                                           "per pair, SimilarityCalculator.py:13-14) takes %.1f ms per pair at N=%d"
                                           % (ns, ns * (ns - 1) // 2, t_cpu, t_lit * 1e3, ns),
                                 "literal_ms_per_pair": t_lit * 1e3,
                                 "drop_in_ms_per_pair": t_drop * 1e3},
                "note": "fp64 in, fp64 / int64 out, every value that reaches the result computed in fp64; the int8 kernel only "
                        "decides which patch is nearest, with a rigorous error bound, and hands undecided cases to fp64 "
                        "(DESIGN.md 4.4; DLC_SIM_FORCE_F64 in the call's flags runs the fp64 Gram form: the same matrix, 38.9 ms)",
                "max_rel_err_vs_oracle": err})
    del mf, mi, sub, ref, dsn
    out[-1]["direct_evaluations"] = sim_direct(desc)[0]

    # ---- f-4 with the REFERENCE's similarity: frames arriving in batches, each scored against every older resident frame
    # (SimilarityCalculator.similarity_score per pair; create_similarity_matrix.py:34-38 as a robot would run it) ----------
    sb, sk, sex = 32, 5, 30
    def sdav_stream():
        det_ = dlc.SdavLoopClosureDetector(score_s, patches=P, width=H, k=sk, exclusion=sex, capacity=N)
        outs_ = [det_.query_and_insert(desc[lo:lo + sb]) for lo in range(0, N, sb)]
        return det_, torch.cat([o[0] for o in outs_]), torch.cat([o[1] for o in outs_])
    score_s = eng.distinctive_score(desc, 0.5, 0.2)
    ss_ms, _, _, (sdet, sds, sdi) = _timed_path(eng, sdav_stream, reps=2)

    def sdav_stream_piped():                                      # two batches in flight (submit / result, one ticket behind): a
        det_ = dlc.SdavLoopClosureDetector(score_s, patches=P, width=H, k=sk, exclusion=sex, capacity=N)   # batch's small kernels
        outs_, prev_ = [], None                                   # run beside its neighbours' product kernels on a second stream
        for lo in range(0, N, sb):
            t_ = det_.submit(desc[lo:lo + sb])
            if prev_ is not None:
                outs_.append(det_.result(prev_))
            prev_ = t_
        outs_.append(det_.result(prev_))
        return torch.cat([o[0] for o in outs_]), torch.cat([o[1] for o in outs_])
    sp_ms, _, _, (sps, spi) = _timed_path(eng, sdav_stream_piped, reps=2)
    piped_same = bool(torch.equal(spi, sdi) and torch.equal(torch.nan_to_num(sps, posinf=1e300, neginf=-1e300),
                                                            torch.nan_to_num(sds, posinf=1e300, neginf=-1e300)))
    del sps, spi

    def sdav_stream_single():                                     # one frame at a time: the single-query kernels
        det_ = dlc.SdavLoopClosureDetector(score_s, patches=P, width=H, k=sk, exclusion=sex, capacity=N)
        outs_ = [det_.query_and_insert(desc[lo:lo + 1]) for lo in range(N)]
        return torch.cat([o[1] for o in outs_])
    ss1_ms, _, _, sdi1 = _timed_path(eng, sdav_stream_single, reps=1)
    # every row of the stream is the matrix call's column: the detector's ranking of it against a stable sort of that column
    mcol = eng.sdav_similarity_matrix(desc, score_s, 10.0, -10.0, want_int64=False)[0].cpu().numpy()
    agree, checked = 0, 0
    for f_ in range(0, N, max(1, N // 40)):
        nsee = f_ - sex
        if nsee <= 0:
            continue
        col = mcol[:nsee, f_]
        order = np.lexsort((np.arange(nsee), -col))[:sk]
        got_ = sdi[f_, :len(order)].cpu().numpy()
        agree += int(np.array_equal(got_, order)); checked += 1
    nsm = min(N, 10)
    dsm = desc[:nsm].cpu().numpy()
    scm = score_s.cpu().numpy()
    t0 = time.perf_counter()
    for f_ in range(1, nsm):
        for j_ in range(f_):
            with np.errstate(divide="ignore"):
                np.sum(10 - 10 * np.log(osim.weighted_distances(dsm[j_], dsm[f_], osim.match_features(dsm[j_], dsm[f_]), scm)))
    t_cpu = time.perf_counter() - t0
    # every (older frame, new frame) pair's patch products once: the matrix call's int8 work (i8_ops above), spread over the batches
    out.append({"path": "SdavLoopClosureDetector.query_and_insert (batches of %d frames)" % sb,
                "reference": "src/sdav/similarity/SimilarityCalculator.py:12-49 per arriving frame (src/sdav/create_similarity_matrix.py:34-38)",
                "frames": N, "dtype": "f64", "k": sk, "value": (N * (N - 1) / 2.0) / (ss_ms * 1e-3), "unit": "frame-pairs/s", "ms": ss_ms,
                "frames_per_s": N / (ss_ms * 1e-3),
                "roofline": {"bound": "mfma", "achieved": i8_ops / (ss_ms * 1e-3) / 1e12, "peak": MFMA_I8_PEAK_TOPS, "unit": "TOP/s",
                             "frac": i8_ops / (ss_ms * 1e-3) / 1e12 / MFMA_I8_PEAK_TOPS, "traffic": None,
                             "kernel": "per batch: the quantisation of the batch into the resident panel, gram_i8_kernel on the strip "
                                       "(columns: the batch's frames, rows: every older patch) + strip_resolve_kernel + stream_score_kernel + "
                                       "topk_rows_f64_kernel -- the all-vs-all call's products, one strip of column tiles per batch",
                             "kernel_ms": ss_ms, "call_ms": ss_ms, "algorithmic_ops_per_call": i8_ops},
                "cpu_baseline": {"value": (nsm * (nsm - 1) // 2) / t_cpu, "unit": "frame-pairs/s", "cores": cores, "kind": "port",
                                 "sample": "oracle/similarity.py similarity_score terms for every frame of the first %d against its older "
                                           "frames (%d pairs): %.2f s" % (nsm, nsm * (nsm - 1) // 2, t_cpu)},
                "ranking_equals_matrix_columns": agree == checked, "frames_checked": checked,
                "stream_poisoned": int(sdet.stream.stats[1]),
                "two_batches_in_flight": {"ms": sp_ms, "frames_per_s": N / (sp_ms * 1e-3), "frac": i8_ops / (sp_ms * 1e-3) / 1e12 / MFMA_I8_PEAK_TOPS,
                                          "same_lists_as_batch_by_batch": piped_same,
                                          "what": "SdavLoopClosureDetector.submit / result: copy + quantisation of batch b beside "
                                                  "batch b - 1's product kernel, resolution + scores + ranking of b - 1 beside b's"},
                "one_frame_at_a_time": {"ms": ss1_ms, "us_per_frame": ss1_ms * 1e3 / N, "frames_per_s": N / (ss1_ms * 1e-3),
                                        "same_lists_as_batched": bool(torch.equal(sdi1, sdi))}})
    # ... and with two batches in flight, a row of its own (the same work, the same lists)
    out.append({"path": "SdavLoopClosureDetector.submit / result (batches of %d frames, two in flight)" % sb,
                "reference": out[-1]["reference"], "frames": N, "dtype": "f64", "k": sk,
                "value": (N * (N - 1) / 2.0) / (sp_ms * 1e-3), "unit": "frame-pairs/s", "ms": sp_ms, "frames_per_s": N / (sp_ms * 1e-3),
                "roofline": {"bound": "mfma", "achieved": i8_ops / (sp_ms * 1e-3) / 1e12, "peak": MFMA_I8_PEAK_TOPS, "unit": "TOP/s",
                             "frac": i8_ops / (sp_ms * 1e-3) / 1e12 / MFMA_I8_PEAK_TOPS, "traffic": None,
                             "kernel": "gram_i8_kernel on the strips, alone on the engine's second stream; a batch's copy + quantisation "
                                       "and its neighbour's strip_resolve_kernel + stream_score_kernel + topk_rows_f64_kernel beside it "
                                       "on the caller's stream", "kernel_ms": sp_ms, "call_ms": sp_ms, "algorithmic_ops_per_call": i8_ops},
                "cpu_baseline": out[-1]["cpu_baseline"], "same_lists_as_batch_by_batch": piped_same,
                "batch_by_batch_ms": ss_ms})
    del sdet, sds, sdi, sdi1, mcol, dsm

    # ---- M1/M2 on real-image statistics: the repo's 20 real frames (tests/golden) tiled to N, through the GPU front-end and
    # SDAV.transform with the reference's N(0,1) initialiser (real images saturate it) and with 1/sqrt(fan_in) weights
    # (every descriptor column within 1e-3 of its own mean, the means spread over [0.15, 0.88]: low contrast) -------------
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import real_frames
    xr = real_frames.tiled_patches(dlc, N)
    for scale, label in (("reference", "N(0,1) weights"), ("fan_in", "1/sqrt(fan_in) weights")):
        dr = dlc.SDAV(seed=4, weight_scale=scale).transform_tensor(xr).reshape(N, P, H)

        def sim_r():
            score, rng = eng.distinctive_score(dr, 0.5, 0.2, with_range=True)
            return eng.sdav_similarity_matrix(dr, score, 10.0, -10.0, range=rng)
        r_ms, rk_ms, rk_n, (rf, ri) = _timed_path(eng, sim_r, reps=4)
        rf, ri = rf.clone(), ri.clone()
        direct, why = sim_direct(dr)
        f64_ms, _, _, (ff, fi) = _timed_path(eng, lambda: eng.sdav_similarity_matrix(dr, eng.distinctive_score(dr, 0.5, 0.2), 10.0, -10.0,
                                                                                      force_f64=True), reps=2)
        same = bool(torch.equal(torch.nan_to_num(rf, posinf=1e300), torch.nan_to_num(ff, posinf=1e300)) and torch.equal(ri, fi))
        nr = min(N, 12)                                            # the oracle's all-vs-all loop on the first frames of THIS data
        drn = dr[:nr].cpu().numpy()
        t0 = time.perf_counter()
        with np.errstate(divide="ignore"):
            osim.similarity_matrix_f64(drn)
        tr_cpu = time.perf_counter() - t0
        out.append({"path": "SDAV similarity matrix, real-frame statistics, " + label,
                    "reference": "src/sdav/similarity/SimilarityCalculator.py:12-49, src/sdav/create_similarity_matrix.py:23-38",
                    "frames": N, "dtype": "f64", "value": pairs / (r_ms * 1e-3), "unit": "frame-pairs/s", "ms": r_ms,
                    "data": "tests/golden: the reference's 20 datasets/test frames tiled to %d (exact copies, copies with pixels "
                            "moved by 1/255, blank and repeated patches), GPU front-end + SDAV.transform" % N,
                    "stats": [direct, why], "direct_evaluations": direct, "arg_mins": pairs * P,
                    "filter_took_the_call": why == 0, "fp64_route_ms": f64_ms, "equals_fp64_route_bit_for_bit": same,
                    "roofline": {"bound": "mfma", "achieved": i8_ops / (rk_ms * 1e-3) / 1e12 if rk_ms else None, "peak": MFMA_I8_PEAK_TOPS,
                                 "unit": "TOP/s", "frac": i8_ops / (rk_ms * 1e-3) / 1e12 / MFMA_I8_PEAK_TOPS if rk_ms else None,
                                 "traffic": None, "kernel": "gram_i8_kernel", "kernel_ms": rk_ms, "kernel_launches_timed": rk_n,
                                 "call_ms": r_ms, "algorithmic_ops_per_call": i8_ops},
                    "cpu_baseline": {"value": (nr * (nr - 1) // 2) / tr_cpu, "unit": "frame-pairs/s", "cores": cores, "kind": "port",
                                     "sample": "oracle/similarity.py all-vs-all loop on the first %d of these frames (%d pairs): %.2f s"
                                               % (nr, nr * (nr - 1) // 2, tr_cpu)}})
        del dr, rf, ri, ff, fi
    del xr

    # ---- M5 at configs[1]: cosine matrix and top-20 over the flattened 75 000-d SDAV place descriptors --------
    db = dlc.KeyframeDatabase(h.reshape(N, P * H), dtype="bf16", center=True)
    rows = db.rows
    d_st = rows.shape[1]
    call_ms, _, _, sm = _timed_path(eng, lambda: eng.cosine_scores(rows, rows))
    cflops = 2.0 * N * N * d_st
    ns = min(N, 256)
    rh = rows.float().cpu().numpy().astype(np.float64)
    t0 = time.perf_counter()
    ref = ocos.scores(rh[:ns], rh)
    t_cpu = time.perf_counter() - t0
    err = float(np.abs(sm[:ns].cpu().numpy() - ref).max())
    tf = cflops / (call_ms * 1e-3) / 1e12
    out.append({"path": "cosine similarity matrix (flattened SDAV descriptors)", "reference": "BASELINE.json configs[1] "
                "(no reference implementation: SURVEY 8a-M5)", "frames": N, "dim": d_st, "dtype": "bf16",
                "value": N * N / (call_ms * 1e-3), "unit": "frame-pairs/s", "ms": call_ms,
                "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": tf / MFMA_PEAK_TFLOPS, "traffic": None, "kernel": "score_gemm_kernel (split-K partial "
                             "tiles) + splitk_dense_kernel", "kernel_ms": call_ms, "call_ms": call_ms,
                             "algorithmic_flops_per_call": cflops},
                "cpu_baseline": {"value": ns * N / t_cpu, "unit": "frame-pairs/s", "cores": cores, "kind": "port",
                                 "sample": "oracle/cosine.py scores (fp64 NumPy matmul) of the first %d frames against all %d: "
                                           "%.2f s" % (ns, N, t_cpu)},
                "max_abs_err_vs_oracle": err})
    call_ms, _, _, top = _timed_path(eng, lambda: eng.match_topk(rows, rows, 20, details=True))
    ti = top.idx
    t0 = time.perf_counter()
    es, ei = ocos.topk_from_scores(ref, 20)
    t_cpu += time.perf_counter() - t0
    tf = cflops / (call_ms * 1e-3) / 1e12
    out.append({"path": "cosine top-20 (flattened SDAV descriptors)", "reference": "BASELINE.json configs[1] / north_star "
                "(no reference implementation)", "frames": N, "dim": d_st, "dtype": "bf16", "k": 20,
                "value": N / (call_ms * 1e-3), "unit": "query-frames/s", "ms": call_ms,
                "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": tf / MFMA_PEAK_TFLOPS, "traffic": None, "kernel": "score_gemm_kernel (split-K) + "
                             "splitk_groups_kernel + finish_topk_kernel (small-database plan: fp64 re-score of the k + 4 best "
                             "rows per query) + exhaustive_topk_kernel", "kernel_ms": call_ms,
                             "call_ms": call_ms, "algorithmic_flops_per_call": cflops},
                "cpu_baseline": {"value": ns / t_cpu, "unit": "query-frames/s", "cores": cores, "kind": "port",
                                 "sample": "oracle/cosine.py scores + exact top-20 for the first %d frames: %.2f s" % (ns, t_cpu)},
                # crowded scores (untrained encoder, random frames): the order is decided on fp64 scores; queries whose
                # k-th score the certificate cannot clear are resolved by the exhaustive pass inside the same call
                "topk_index_agreement_vs_oracle": float((ti[:ns].cpu().numpy() == ei).mean()),
                "topk_score_max_abs_err_vs_oracle": float(np.abs(top.scores_f64[:ns].cpu().numpy() - es).max()),
                "queries_resolved_by_exhaustive_pass": int((top.status == 2).sum())})
    ts = None
    del db, rows, sm, rh, ref, ts, ti, h, desc, x

    # ---- E8-E11: CnnVtl.transform (cnn_vtl.py:28-133) on 192x240 frames (configs[2]) -----------------------
    frames = torch.randint(0, 256, (N, 192, 240, 3), generator=g, device=eng.device).to(torch.float64)
    cnn = dlc.CnnVtl(input_shape=[N, 192, 240, 3], seed=3, mask_seed=4)
    call_ms, k_ms, k_n, d8 = _timed_path(eng, lambda: cnn.transform_tensor(frames), reps=2)
    flops, hh, ww = 0.0, 192, 240
    for (kh, kw, cin, cout, s_, ph, pw, oh, ow, relu, pool) in cnn._geom:
        flops += 2.0 * N * oh * ow * kh * kw * cin * cout
    nb = min(N, 6)
    cw, cb = ocnn.init_weights(3)
    fh = frames[:nb].cpu().numpy()
    t0 = time.perf_counter()
    ref = ocnn.transform(fh, cw, cb, ocnn.column_indices(cnn.layer_sizes, 99.59, seed=4))
    t_cpu = time.perf_counter() - t0
    diff = int((d8[:nb].cpu().numpy() != ref).sum())
    # ndarray in, ndarray out: uint8 frames as cv2.imread hands them over (create_distance_matrix.py:23), and float64
    f8 = frames.to(torch.uint8).cpu().numpy()
    h2h_ms, d_np = _timed_host(lambda: cnn.transform(f8))
    h2h_same = bool(np.array_equal(d_np, d8.cpu().numpy()))
    f64_np = frames.cpu().numpy()
    h2h64_ms, _ = _timed_host(lambda: cnn.transform(f64_np), reps=2)
    del f8, f64_np, d_np
    out.append({"path": "CnnVtl.transform", "reference": "src/cnn_vtl/network/cnn_vtl.py:28-133", "frames": N, "dtype": "f64",
                "descriptor_bytes": int(d8.shape[1]), "value": N / (call_ms * 1e-3), "unit": "frames/s", "ms": call_ms,
                "host_to_host_ms": h2h_ms, "host_to_host_ms_float64_frames": h2h64_ms, "host_to_host_bit_identical": h2h_same,
                "roofline": _mfma_f64_roofline(flops, k_ms, k_n, call_ms, "gemm_dma_f64_kernel (implicit-GEMM conv1..conv5, fused bias + ReLU)"),
                "cpu_baseline": {"value": nb / t_cpu, "unit": "frames/s", "cores": cores, "kind": "port",
                                 "sample": "oracle/cnn_vtl.py (fp64 NumPy im2col + matmul) on the first %d frames, same weights and "
                                           "columns: %.1f s" % (nb, t_cpu)},
                "int8_bytes_differing_from_oracle": diff, "int8_bytes_compared": int(ref.size)})
    del frames, fh, ref, cw, cb

    # ---- M3/M4: cnn_vtl distance matrix (DistanceCalculator.py:4-12 + create_distance_matrix.py:30-36) -------
    desc8 = d8.contiguous()
    dp = int(desc8.shape[1])
    call_ms, _, _, dm = _timed_path(eng, lambda: eng.cnnvtl_distance_matrix(desc8))
    bytes_alg = N * dp + N * N * 8                                  # descriptors read once + the int64 matrix written
    ns = min(N, 160)
    dh = desc8[:ns].cpu().numpy()
    t0 = time.perf_counter()
    ref = odist.distance_matrix(dh)
    t_cpu = time.perf_counter() - t0
    npair = min(ns * ns, 1500)                                      # the literal form: one Python-level call per pair
    t0 = time.perf_counter()
    for e in range(npair):
        odist.calculate_distance(dh[e // ns], dh[e % ns])
    t_lit = (time.perf_counter() - t0) / npair
    dlc.DistanceCalculator.calculate_distance(dh[0], dh[1])         # the drop-in class, one call per pair
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for e in range(200):
        dlc.DistanceCalculator.calculate_distance(dh[e % ns], dh[(e + 1) % ns])
    t_drop = (time.perf_counter() - t0) / 200
    exact = bool(np.array_equal(dm[:ns, :ns].cpu().numpy(), ref))
    gbs = bytes_alg / (call_ms * 1e-3) / 1e9
    out.append({"path": "cnn_vtl distance matrix", "reference": "src/cnn_vtl/similarity/DistanceCalculator.py:4-12, "
                "src/cnn_vtl/create_distance_matrix.py:30-36", "frames": N, "descriptor_bytes": dp, "dtype": "i8",
                "value": N * N / (call_ms * 1e-3), "unit": "frame-pairs/s", "ms": call_ms,
                "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                             "traffic": None, "kernel": "distance_matrix_kernel", "kernel_ms": call_ms, "call_ms": call_ms,
                             "algorithmic_bytes_per_call": bytes_alg,
                             "byte_pairs_per_s": N * N * dp / (call_ms * 1e-3),
                             "valu_frac": (N * (N + 1) / 2.0) * ((dp + 3) // 4) * 8 / (call_ms * 1e-3) / 39.3e12,
                             "note": "N*D' bytes in, N*N*8 out: the kernel is VALU work (xor, |x|, popcount on N*N*D' byte "
                                     "pairs), not HBM traffic; byte_pairs_per_s is its own rate; valu_frac = 8 integer "
                                     "instructions per 4-byte word pair of the upper triangle against 39.3 T lane-operations/s "
                                     "(256 CUs x 64 lanes x 2.4 GHz): 0.3 at 1063 frames (153 tiles of 64 x 64 frames: a small "
                                     "launch), 0.66 at 4000 frames (scripts/exp_distance.py)"},
                "cpu_baseline": {"value": ns * ns / t_cpu, "unit": "frame-pairs/s", "cores": 1, "kind": "port",
                                 "sample": "oracle/distance.py (NumPy table lookup per row) on %d x %d frames: %.2f s; one "
                                           "calculate_distance call per pair, the reference's loop shape: %.3f ms per pair"
                                           % (ns, ns, t_cpu, t_lit * 1e3), "literal_ms_per_pair": t_lit * 1e3,
                                 "drop_in_ms_per_pair": t_drop * 1e3},
                "bit_exact_vs_oracle": exact})
    del desc8, dm, d8, cnn
    out.extend(bench_end_to_end(eng, dlc, N, {p_["path"]: p_ for p_ in out}))
    return out


def _stage_roofline(stages, call_ms, candidates):
    """Roofline entry of a composed path: its dominant stage (of `candidates`: (stage name, algorithmic operations, peak in
    T-ops/s, unit, kernels)) timed by HIP events inside the call."""
    name, ops, peak, unit, kern = max(candidates, key=lambda c: stages.get(c[0], 0.0))
    ms = stages[name]
    call_ms = max(call_ms, sum(stages.values()))               # (the stages were timed in a call of their own)
    ach = ops / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak, "traffic": None,
            "kernel": "dominant stage of the call: %s -- %s" % (name, kern), "kernel_ms": ms, "call_ms": call_ms,
            "dominant_stage_share_of_call": ms / call_ms, "algorithmic_ops_of_the_stage": ops}


def bench_end_to_end(eng, dlc, N, stage_rows):
    """BASELINE configs[0], configs[1] and configs[2] as ONE timed path each -- what the reference's user runs
    (src/sdav/create_similarity_matrix.py:23-38, src/cnn_vtl/create_distance_matrix.py:14-36): frames in, matrix out, through
    deeploopcloser_amd/pipeline.py with no host hop between the stages.  Frames: the repo's 20 real frames tiled to N
    (tests/real_frames.py).  Each row: device-resident ms (uint8 frames already in HBM -> the matrix in HBM), host-to-host
    ms (ndarray in -> ndarray out: one chunked upload overlapping the first kernels, one download), the stage breakdown
    (HIP events around every stage of every chunk), the sum of the separately timed stage rows above for comparison, the
    matrix checked bit for bit against the staged calls, and the CPU oracle's end-to-end time on a 20-frame sample."""
    import gc
    from deeploopcloser_amd import pipeline
    from oracle import similarity as osim, distance as odist, cosine as ocos
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import real_frames
    import config1_common as c1
    gc.collect()
    P, H = 30, 2500
    cores = blas_threads()
    rows = []
    frames = real_frames.tiled_u8_frames(dlc, N)                           # host, uint8 RGB
    frames_dev = torch.from_numpy(frames).to(eng.device)
    parser = dlc.CvInputParser(P, 41)

    def timed_host(fn, reps=3):
        return _timed_host(fn, reps=reps)

    def wall_dev(fn, reps=3):
        fn(); torch.cuda.synchronize()
        best = None
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); r = fn(); e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1)
            best = (t, r) if best is None or t < best[0] else best
        return best

    # the CPU oracle end to end on a 20-frame sample (the reference's datasets/test size): patches, encoder, matrix
    paths20 = c1.frame_paths()[:min(20, N)]
    t0 = time.perf_counter()
    x20 = c1.oracle_patches(paths20)
    t_pat = time.perf_counter() - t0
    ns = len(paths20)

    for mode, label in (("float64", "fp64 encoder (parity mode)"), ("f16x2", "f16x2 encoder (tolerance mode)")):
        net = dlc.SDAV(seed=4, weight_scale="fan_in", dtype=mode)
        net.transform_tensor(torch.zeros((2, P, 1681), dtype=torch.float64, device=eng.device))     # (weight panels prepared once)
        dev_ms, m_dev = wall_dev(lambda: pipeline.sdav_similarity_matrix_from_frames(frames_dev, net, parser, device_result=True))
        m_dev = m_dev.clone()
        h2h_ms, m_host = timed_host(lambda: pipeline.sdav_similarity_matrix_from_frames(frames, net, parser))
        tm = []
        pipeline.sdav_similarity_matrix_from_frames(frames, net, parser, timings=tm)
        stages = pipeline.stage_ms(tm)
        tm = []
        pipeline.sdav_similarity_matrix_from_frames(frames_dev, net, parser, device_result=True, timings=tm)
        stages_dev = pipeline.stage_ms(tm)
        # the staged calls (each stage on its own, results through device tensors): the same matrix, bit for bit
        x = parser.parse_batch(frames_dev)
        desc = net.transform_tensor(x).view(N, P, H)
        sc, rg = eng.distinctive_score(desc, 0.5, 0.2, with_range=True)
        want = eng.sdav_similarity_matrix(desc, sc, 10.0, -10.0, range=rg)[1]
        same = bool(torch.equal(want, m_dev) and np.array_equal(m_host, want.cpu().numpy()))
        del x, desc, want
        # oracle: encoder + all-vs-all similarity on the 20-frame sample, same weights
        ws, bs = net.get_weights()
        from oracle import sdav as osdav
        t0 = time.perf_counter()
        h20 = osdav.transform(x20, ws, bs).reshape(ns, P, H)
        with np.errstate(divide="ignore"):
            ref20 = osim.similarity_matrix(h20)
        t_cpu = t_pat + (time.perf_counter() - t0)
        got20 = pipeline.sdav_similarity_matrix_from_frames(frames[:ns], net, parser)
        if mode == "float64":
            # the oracle's patches are the front-end's bit for bit (tests): a truncated score may differ only where it sits on an integer
            agree20 = float((got20 == ref20).mean())
        else:
            agree20 = None
        stage_sum = None
        enc_row = stage_rows.get("SDAV.transform" if mode == "float64" else "SDAV.transform (f16x2 split, tolerance mode)")
        fe_row = stage_rows.get("patch front-end (grey + Harris + %d patches of 41x41)" % P)
        sim_row = stage_rows.get("SDAV similarity matrix, real-frame statistics, 1/sqrt(fan_in) weights")
        if enc_row and fe_row and sim_row:
            stage_sum = fe_row["ms"] + enc_row["ms"] + sim_row["ms"]
        pairs = N * (N - 1) // 2
        rows.append({"path": "configs[1] end to end: %d frames -> patches -> SDAV -> similarity matrix, %s" % (N, label),
                     "reference": "src/sdav/create_similarity_matrix.py:23-38 (CvInputParser.py:19-49, SDAV.py:293-302, "
                                  "SimilarityCalculator.py:12-49)",
                     "frames": N, "dtype": "f64" if mode == "float64" else "f16x2 encode, f64 similarity",
                     "data": "tests/golden: the reference's 20 datasets/test frames tiled to %d (uint8 RGB, 192 x 240)" % N,
                     "value": N / (dev_ms * 1e-3), "unit": "frames/s", "ms": dev_ms,
                     "device_resident_ms": dev_ms, "host_to_host_ms": h2h_ms,
                     "stage_ms_device_resident": stages_dev, "stage_ms_inside_the_host_to_host_call": stages,
                     "sum_of_the_separately_timed_stage_rows_ms": stage_sum,
                     "end_to_end_over_stage_sum": dev_ms / stage_sum if stage_sum else None,
                     "equals_staged_calls_bit_for_bit": same,
                     # the roofline of the path's dominant STAGE inside this call (each stage's kernel has its own row above)
                     "roofline": _stage_roofline(stages_dev, dev_ms, [
                         ("SDAV.transform", (1.0 if mode == "float64" else 3.0) * 2.0 * P * N * (1681 * H + 4 * H * H),
                          MFMA_F64_PEAK_TFLOPS if mode == "float64" else MFMA_PEAK_TFLOPS, "TFLOP/s",
                          "gemm_dma_f64_kernel (5 layers)" if mode == "float64" else "gemm_split_f16_kernel (5 layers x 3 fp16 products)"),
                         ("similarity matrix (distinctive score + all-vs-all)", 6 * 2.0 * (N * (N - 1) / 2.0) * P * P * H,
                          MFMA_I8_PEAK_TOPS, "TOP/s", "gram_i8_kernel + distinctive score + resolution")]),
                     "cpu_baseline": {"value": ns / t_cpu, "unit": "frames/s", "cores": cores, "kind": "port",
                                      "sample": "oracle end to end on the first %d real frames (oracle/patches.py + keypoints.py %.1f s, "
                                                "oracle/sdav.py + oracle/similarity.py all-vs-all: %.1f s in all); the cost grows with the "
                                                "square of the frame count (%d pairs here, %d at %d frames)"
                                                % (ns, t_pat, t_cpu, ns * (ns - 1) // 2, pairs, N)},
                     "matrix_agreement_with_oracle_on_the_sample": agree20})
        del net, m_dev, m_host

    # ---- configs[2]
    bgr = np.ascontiguousarray(frames[..., ::-1])                         # cv2.imread order (create_distance_matrix.py:23)
    bgr_dev = torch.from_numpy(bgr).to(eng.device)
    cnn = dlc.CnnVtl(input_shape=[N, 192, 240, 3], seed=3, mask_seed=4)
    dev_ms, m_dev = wall_dev(lambda: pipeline.cnn_vtl_distance_matrix_from_frames(bgr_dev, cnn, device_result=True), reps=2)
    m_dev = m_dev.clone()
    h2h_ms, m_host = timed_host(lambda: pipeline.cnn_vtl_distance_matrix_from_frames(bgr, cnn), reps=2)
    tm = []
    pipeline.cnn_vtl_distance_matrix_from_frames(bgr, cnn, timings=tm)
    stages = pipeline.stage_ms(tm)
    tm = []
    pipeline.cnn_vtl_distance_matrix_from_frames(bgr_dev, cnn, device_result=True, timings=tm)
    stages_dev = pipeline.stage_ms(tm)
    cflops = sum(2.0 * N * oh * ow * kh * kw * cin * cout for (kh, kw, cin, cout, s_, ph, pw, oh, ow, relu, pool) in cnn._geom)
    want = eng.cnnvtl_distance_matrix(cnn.transform_tensor(bgr_dev.to(torch.float64)))
    same = bool(torch.equal(want, m_dev) and np.array_equal(m_host, want.cpu().numpy()))
    nb = min(N, 6)
    from oracle import cnn_vtl as ocnn
    cw, cb = ocnn.init_weights(3)
    t0 = time.perf_counter()
    d_ref = ocnn.transform(bgr[:nb].astype(np.float64), cw, cb, ocnn.column_indices(cnn.layer_sizes, 99.59, seed=4))
    dm_ref = odist.distance_matrix(d_ref)
    t_cpu = time.perf_counter() - t0
    exact = bool(np.array_equal(pipeline.cnn_vtl_distance_matrix_from_frames(bgr[:nb], cnn), dm_ref))
    stage_sum = None
    if stage_rows.get("CnnVtl.transform") and stage_rows.get("cnn_vtl distance matrix"):
        stage_sum = stage_rows["CnnVtl.transform"]["ms"] + stage_rows["cnn_vtl distance matrix"]["ms"]
    rows.append({"path": "configs[2] end to end: %d frames -> CnnVtl -> distance matrix" % N,
                 "reference": "src/cnn_vtl/create_distance_matrix.py:14-36 (cnn_vtl.py:28-133, DistanceCalculator.py:4-12)",
                 "frames": N, "dtype": "f64 convolutions, int8 descriptors, int64 distances",
                 "data": "tests/golden: the reference's 20 datasets/test frames tiled to %d (uint8 BGR, 192 x 240)" % N,
                 "value": N / (dev_ms * 1e-3), "unit": "frames/s", "ms": dev_ms,
                 "device_resident_ms": dev_ms, "host_to_host_ms": h2h_ms, "stage_ms_device_resident": stages_dev,
                 "stage_ms_inside_the_host_to_host_call": stages,
                 "sum_of_the_separately_timed_stage_rows_ms": stage_sum,
                 "end_to_end_over_stage_sum": dev_ms / stage_sum if stage_sum else None,
                 "equals_staged_calls_bit_for_bit": same,
                 "roofline": _stage_roofline(stages_dev, dev_ms, [
                     ("CnnVtl.transform", cflops, MFMA_F64_PEAK_TFLOPS, "TFLOP/s",
                      "uint8 -> fp64, implicit-GEMM conv1..conv5 on gemm_dma_f64_kernel, pooling, min/max, quantise + gather")]),
                 "cpu_baseline": {"value": nb / t_cpu, "unit": "frames/s", "cores": cores, "kind": "port",
                                  "sample": "oracle/cnn_vtl.py + oracle/distance.py end to end on the first %d frames: %.1f s" % (nb, t_cpu)},
                 "bit_exact_vs_oracle_on_the_sample": exact})
    del cnn, m_dev, m_host, bgr_dev, want

    # ---- configs[0]: the reference's own CPU-runnable case -- the 20 frames of datasets/test, SDAV encode + the all-vs-all
    # cosine matrix of the flattened descriptors (BASELINE.json configs[0]; tests/test_config1.py holds the parity side)
    n0 = min(20, N)
    net = dlc.SDAV(seed=c1.SEED, weight_scale="fan_in")
    f0 = frames[:n0]

    def config0():
        d_ = pipeline.sdav_descriptors_from_frames(f0, net, parser)
        st = eng.normalize(d_.view(n0, -1), "bf16", center=True)
        return eng.download(eng.cosine_scores(st, st))
    with eng.latency_mode():                  # 600 rows are 200 tiles of 64 x 128: split-K for the encoder's GEMMs (1.56 -> 1.43 ms,
        c0_ms, cm = timed_host(config0)       # scripts/exp/r06_config0.py; the 20 x 20 bf16 cosines come out the same bits)
    t0 = time.perf_counter()
    h0 = c1.oracle_descriptors(x20[:n0], "fan_in")
    ref0 = c1.oracle_cosine(h0, n0)
    t_cpu0 = t_pat * n0 / max(1, ns) + (time.perf_counter() - t0)
    rows.append({"path": "configs[0] end to end: %d real frames -> patches -> SDAV -> %d x %d cosine matrix" % (n0, n0, n0),
                 "reference": "BASELINE.json configs[0] (src/sdav/network/SDAV.py:293-302 on datasets/test; the cosine is north_star's)",
                 "frames": n0, "dtype": "f64 encode, bf16 cosine", "value": n0 / (c0_ms * 1e-3), "unit": "frames/s", "ms": c0_ms,
                 "host_to_host_ms": c0_ms,
                 "roofline": {"bound": "mfma", "achieved": 2.0 * P * n0 * (1681 * H + 4 * H * H) / (c0_ms * 1e-3) / 1e12,
                              "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": 2.0 * P * n0 * (1681 * H + 4 * H * H) / (c0_ms * 1e-3) / 1e12 / MFMA_F64_PEAK_TFLOPS, "traffic": None,
                              "kernel": "the whole host-to-host call against the encoder's fp64 work: latency-bound at 20 frames (upload, "
                                        "front-end, five 600-row fp64 GEMMs in the engine's latency mode (split-K), normalise, one split-K "
                                        "score pass, download)",
                              "kernel_ms": c0_ms, "call_ms": c0_ms},
                 "cpu_baseline": {"value": n0 / t_cpu0, "unit": "frames/s", "cores": cores, "kind": "port",
                                  "sample": "the CPU oracle on the same %d frames (patches + oracle/sdav.py + oracle/cosine.py): %.1f s" % (n0, t_cpu0)},
                 "max_abs_err_vs_oracle": float(np.abs(cm - ref0).max())})
    return rows


def bench_shard_emulation(eng, dlc, rows, queries, k, want_idx, steps=100, ranks=(2, 4, 8)):
    """What ONE rank of an R-GPU run does per query batch, on this one GPU: MatchPipeline's sharded protocol (score pass,
    group selection, all-gather #1, filtered fp64 re-score, all-gather #2, certifying merge; three batches in flight, two
    streams) over rank 0's shard of the resident database.  The whole database is on this GPU, so the OTHER ranks'
    contributions are real: their group maxima and -- filtered against everybody's maxima, as the protocol has it -- their
    packed top-k parts are computed once, before anything is timed, from their shards; the two all-gathers are then device
    copies of this rank's fresh part next to those.  NO RCCL, no other GPU: the GPU-side floor of a rank's step, i.e. an upper
    bound of what R GPUs reach (every rank in lockstep); the merged result must equal the one-GPU result (want_idx)."""
    out = []
    n, nq, d = rows.shape[0], queries.shape[0], queries.shape[1]
    kg = eng.groups_per_query(k)
    for parts in ranks:
        bounds = [dlc.shard_bounds(n, parts, r) for r in range(parts)]
        gmx, ids = [], []
        for lo, hi in bounds:                                   # step 1 of every rank: its kg best groups
            ws = torch.empty(eng.topk_workspace_bytes(nq, hi - lo, d, k), dtype=torch.uint8, device=eng.device)
            gi = torch.empty((nq, kg), dtype=torch.int32, device=eng.device)
            gm = torch.empty((nq, kg + 1), dtype=torch.float32, device=eng.device)
            eng.score_groups(queries, rows[lo:hi], k, ws)
            eng.select_groups(queries, rows[lo:hi], k, ws, gi, gm)
            gmx.append(gm); ids.append(gi)
            del ws
        all_max = torch.stack(gmx)                              # what all-gather #1 delivers
        packs = []
        for r, (lo, hi) in enumerate(bounds):                   # step 2 of every rank: its filtered fp64 part, packed
            pack = torch.empty(nq * k * 16, dtype=torch.uint8, device=eng.device)
            p_idx = pack[:nq * k * 8].view(torch.int64).view(nq, k)
            p_s64 = pack[nq * k * 8:].view(torch.float64).view(nq, k)
            eng.rescore_topk(queries, rows[lo:hi], k, ids[r], gmx[r], p_s64, p_idx, all_max=all_max, row_offset=lo)
            packs.append(pack)
        others_max = all_max[1:].reshape(parts - 1, -1).clone()
        others_pack = torch.stack(packs[1:])
        torch.cuda.synchronize()

        def fake_all_gather(out_t, inp, group=None, others_max=others_max, others_pack=others_pack, parts=parts):
            o = out_t.view(parts, -1)
            o[0].copy_(inp.reshape(-1))
            o[1:].copy_(others_max if inp.dtype == torch.float32 else others_pack)

        db = dlc.KeyframeDatabase(rows[bounds[0][0]:bounds[0][1]], dtype=rows.dtype, stored=True)
        pipe = dlc.MatchPipeline(db, k, depth=3)
        pipe.world = parts                                     # the sharded branch of submit() / result()
        pipe.all_gather = fake_all_gather
        t = None
        for _ in range(10):
            t = pipe.submit(queries)
        pipe.result(t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            t = pipe.submit(queries)
            if t >= pipe.depth - 1:
                pipe.result(t - (pipe.depth - 1))
        s_e, i_e = pipe.result(t)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        out.append({"ranks": parts, "rows_per_rank": int(bounds[0][1] - bounds[0][0]), "ms_per_batch_per_rank": ms,
                    "projected_query_frames_per_s": nq / (ms * 1e-3), "resolved_batches": pipe.resolved_batches,
                    "merged_result_equals_one_gpu": bool(torch.equal(i_e, want_idx)),
                    "label": "EMULATED on one GPU: rank 0's MatchPipeline step over N / %d rows; the other ranks' parts are real "
                             "(computed once from their shards), the two all-gathers are device copies -- no RCCL, no xGMI; an "
                             "upper bound for %d GPUs" % (parts, parts)})
        del pipe, db, packs, others_pack, all_max
    return out


def _timed_match(eng, db, q, k, steps=20, warmup=3):
    """steps one-shot top-k matches of the stored queries q against the resident database, the headline's harness: wall
    clock between two fences, the score pass's launches from the library's HIP events.  Returns (ms per step, mean ms of
    the score pass, its launches timed, the last result with details)."""
    for _ in range(warmup):
        db.match_topk(q, k)
    eng.set_profiling(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r = db.match_topk(q, k)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    g = eng.profile_gemm_ms(min(steps, 256))
    eng.set_profiling(False)
    top = db.match_topk(q, k, details=True)
    return ms, (float(np.mean(g)) if g else None), len(g), top


def bench_configs(eng, dlc, args, rows_bf16, queries_bf16, planted, noise, sigma, planted_rows):
    """BASELINE.json configs[3] (100 k x 4096, Q = 256, sharded over 8 GPUs) and configs[4] (1 M x 4096 fp16, top-20, 8 GPUs)
    as timed rows on this ONE GPU: the whole database in one shot (what one GPU does alone), and rank 0's MatchPipeline
    step of the 8-GPU form with the other ranks' real parts (bench_shard_emulation: collectives replaced by device copies
    -- an upper bound, labelled so).  Each row: ms per batch of Q queries, query-frames/s, the score pass's roofline, the
    top-k digest (equal between the one-shot and the merged emulated result), recall@1 of the planted neighbours."""
    import hashlib
    n, d, nq, k = args.rows, args.dim, args.queries, args.k
    out = []

    def row(name, cfg, db, q, dtype_name, planted_in_db):
        nrows = len(db)
        ms, g_ms, g_n, top = _timed_match(eng, db, q, k)
        algo = nrows * d * 2 + nq * d * 2
        flops = 2.0 * nq * nrows * d
        kms = g_ms if g_ms else ms
        hit = (top.idx[:, 0].cpu().numpy() == planted_rows)[planted_in_db]
        r = {"config": cfg, "path": name, "db_rows": nrows, "dim": d, "queries": nq, "k": k, "dtype": dtype_name,
             "value": nq / (ms * 1e-3), "unit": "query-frames/s", "ms_per_step": ms, "steps": 20,
             "recall_at_1_of_planted_rows_in_this_db": float(hit.mean()) if hit.size else None,
             "queries_resolved_by_exhaustive_pass": int((top.status == 2).sum()),
             "topk_idx_sha256": hashlib.sha256(np.ascontiguousarray(top.idx.cpu().numpy()).tobytes()).hexdigest(),
             "roofline": {"bound": "hbm", "achieved": algo / (kms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": algo / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, "frac_of_measured_copy": algo / (kms * 1e-3) / 1e9 / HBM_COPY_GBS,
                          "traffic": None, "kernel": "score_gemm_kernel (the plan's score pass)", "kernel_ms": kms,
                          "kernel_launches_timed": g_n, "algorithmic_bytes_per_launch": algo,
                          "mfma_achieved_tflops": flops / (kms * 1e-3) / 1e12, "mfma_frac": flops / (kms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS}}
        return r, top

    # configs[3]: the first 100 000 rows of the resident bf16 database (fewer when --rows is smaller)
    n3 = min(100_000, n)
    db3 = dlc.KeyframeDatabase(rows_bf16[:n3], dtype=rows_bf16.dtype, stored=True)
    r3, top3 = row("configs[3]: %d x %d bf16, Q = %d, top-%d, one GPU" % (n3, d, nq, k), 3, db3, queries_bf16, "bf16", planted_rows < n3)
    if n3 >= 8 * 2048:
        emu = bench_shard_emulation(eng, dlc, rows_bf16[:n3], queries_bf16, k, top3.idx, steps=200, ranks=(8,))
        r3["eight_gpu_emulation"] = emu[0]
    out.append(r3)
    del db3, top3

    # configs[4]: the same synthetic database stored as fp16 (same seeds, same planted rows), fp16 queries
    f16 = torch.float16
    rows16, _ = synth_shard(eng, n, d, 0, n, f16, planted_rows)
    q16 = eng.normalize(planted + sigma * noise, f16, center=True)
    db4 = dlc.KeyframeDatabase(rows16, dtype=f16, stored=True)
    r4, top4 = row("configs[4]: %d x %d fp16, Q = %d, top-%d, one GPU" % (n, d, nq, k), 4, db4, q16, "f16", np.ones(len(planted_rows), dtype=bool))
    if n >= 8 * 2048:
        emu = bench_shard_emulation(eng, dlc, rows16, q16, k, top4.idx, steps=100, ranks=(8,))
        r4["eight_gpu_emulation"] = emu[0]
    out.append(r4)
    del db4, rows16, q16, top4
    return out


LINE_LIMIT = 4096            # the driver's parser lost round 5's 28 KB line: the line on stdout stays under this, always
DETAIL_FILE = "bench_detail.json"

_PATH_IDS = [                # (substring of a `paths` row's name, its key in the line's paths_summary), first match wins
    ("4096-wide", "sdav4096_encode_cos"),
    ("f16x2 split", "sdav_encode_f16x2"), ("SDAV.transform", "sdav_encode_f64"), ("train_step", "sdav_train_step"),
    ("patch front-end", "frontend"), ("SdavLoopClosureDetector.submit", "stream_sdav_piped"), ("SdavLoopClosureDetector", "stream_sdav"), ("LoopClosureDetector", "stream_cosine"),
    ("real-frame statistics, N(0,1)", "sdav_sim_real_n01"), ("real-frame statistics, 1/sqrt", "sdav_sim_real_fanin"),
    ("SDAV similarity matrix", "sdav_sim"), ("cosine similarity matrix", "cos_matrix_75k"), ("cosine top-", "cos_topk_75k"),
    ("CnnVtl.transform", "cnnvtl_encode"), ("cnn_vtl distance matrix", "cnnvtl_dist"),
    ("configs[1] end to end", None), ("configs[2] end to end", "cfg2_e2e"), ("configs[0] end to end", "cfg0_e2e"),
]


def path_id(name):
    for sub, key in _PATH_IDS:
        if sub in name:
            if key is None:
                return "cfg1_e2e_f16x2" if "f16x2" in name else "cfg1_e2e_f64"
            return key
    import zlib                                                # an unlisted row: a short, distinct key from its name
    return "%s_%04x" % (re.sub(r"[^a-z0-9]+", "_", name.lower())[:18], zlib.crc32(name.encode()) & 0xffff)


def _r(x, digits=5):
    """Numbers of the line carry `digits` significant digits (the sidecar keeps them all)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (digits, x))
    return x


def compact_line(out, detail=DETAIL_FILE):
    """The ONE line the driver parses, built from the full result `out` (which goes to the sidecar file whole): the
    contract's keys, `roofline` (with the power probe and the energy per step), `cpu_baseline`, the index agreement, and
    one [ms, roofline fraction, bound] triple per other row of the hot path.  Under LINE_LIMIT bytes whatever `out` holds:
    the optional blocks are dropped, last first, if they do not fit."""
    cfg = dict(out["config"])
    cfg["workload"] = "%d x %d %s keyframe DB, Q=%d, top-%d cosine match, %d GPU(s)" % (
        cfg.get("db_rows", 0), cfg.get("dim", 0), out["dtype"], cfg.get("queries_per_step", 0), cfg.get("k", 0), out["n_gpus"])
    line = {k: _r(out[k], 7) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                        "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = cfg
    line["recall_at_1"] = _r(out.get("recall_at_1"))
    ro = out["roofline"]
    r2 = {k: _r(ro.get(k), 6) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms",
                                         "kernel_launches_timed", "algorithmic_bytes_per_launch", "mfma_frac",
                                         "frac_of_measured_copy", "energy_j_per_step")}
    if ro.get("traffic_source"):
        r2["traffic_source"] = str(ro["traffic_source"]).split(" ")[0] + " (replayed PMC passes)"
    pp = ro.get("power_probe")
    if pp:
        r2["power_probe"] = {k: _r(pp.get(k)) for k in ("package_power_w", "sclk_mhz", "samples", "energy_counter_j_per_step")}
    line["roofline"] = r2
    cb = out.get("cpu_baseline")
    if cb:
        c2 = {k: _r(cb.get(k)) for k in ("value", "unit", "cores", "kind", "arithmetic", "value_f32", "value_torch_f32_all_cores",
                                         "host_cpus")}
        c2["sample"] = cb.get("sample_short") or str(cb.get("sample", ""))[:160]
        line["cpu_baseline"] = c2
    for k in ("topk_index_agreement_vs_oracle", "topk_index_agreement_rows", "topk_score_max_abs_err_vs_oracle", "topk_idx_sha256"):
        if k in out:
            line[k] = _r(out[k])
    optional = []                                             # (key, value), most dispensable LAST
    if out.get("steady_state"):
        optional.append(("steady_state", {"value": _r(out["steady_state"]["value"]), "ms_per_step": _r(out["steady_state"]["ms_per_step"])}))
    if out.get("rccl_smoke"):
        sm = out["rccl_smoke"]
        optional.append(("rccl_smoke", {k: _r(sm.get(k)) for k in ("backend", "rccl_version", "ranks", "pipeline_equals_plain_exchange",
                                                                   "first_collective_s")}))
        if isinstance(sm.get("collective_us"), dict):
            optional[-1][1]["collective_us"] = {k: _r(v, 4) for k, v in sm["collective_us"].items()}
    if out.get("rccl_world1_smoke"):
        optional.append(("rccl_world1_smoke", out["rccl_world1_smoke"]))
    if out.get("pipeline"):
        optional.append(("pipeline", out["pipeline"]))
    if out.get("paths"):
        optional.append(("paths_summary", {"_": "[ms, roofline frac, bound]",
                                           **{path_id(p["path"]): [_r(p["ms"], 4), _r(p["roofline"]["frac"], 3), p["roofline"]["bound"]]
                                              for p in out["paths"]}}))
    if out.get("baseline_configs"):
        optional.append(("configs_summary", {"_": "[ms, hbm frac, emulated 8-GPU q-f/s]",
                                             **{"cfg%d" % c["config"]: [_r(c["ms_per_step"], 4), _r(c["roofline"]["frac"], 3),
                                                                       _r((c.get("eight_gpu_emulation") or {}).get("projected_query_frames_per_s"), 4)]
                                                for c in out["baseline_configs"]}}))
    if out.get("multi_gpu_emulation"):
        optional.append(("emulated_ranks_qfps", {"_": "EMULATED on one GPU, no RCCL",
                                                 **{str(e["ranks"]): _r(e["projected_query_frames_per_s"], 4) for e in out["multi_gpu_emulation"]}}))
    for k in ("finish_ms", "step_minus_gemm_ms"):
        if out.get(k) is not None:
            optional.append((k, _r(out[k], 4)))
    line["detail"] = detail
    for key, val in optional:
        line[key] = val
    txt = json.dumps(line, separators=(",", ":"))
    while len(txt) >= LINE_LIMIT and optional:                  # cannot happen with today's rows; never print a line the
        key, _ = optional.pop()                                 # driver cannot parse
        line.pop(key, None)
        line["dropped_for_size"] = line.get("dropped_for_size", []) + [key]
        txt = json.dumps(line, separators=(",", ":"))
    if len(txt) >= LINE_LIMIT:
        raise SystemExit("bench.py: the contract line is %d bytes" % len(txt))
    return txt


def write_detail(out, path=None):
    """The full result -- every row of `paths`, `baseline_configs`, the emulation, the probes -- beside the script (and
    under gpurun_out/ when that exists, which is what travels back from a GPU box).  Never on stdout or stderr."""
    txt = json.dumps(out, indent=1)
    targets = [path or os.path.join(ROOT, DETAIL_FILE)]
    scratch = os.path.join(ROOT, "gpurun_out")
    if path is None and os.path.isdir(scratch):
        targets.append(os.path.join(scratch, DETAIL_FILE))
    for t in targets:
        try:
            with open(t, "w") as f:
                f.write(txt + "\n")
        except OSError as e:                                     # a read-only checkout must not cost the line
            print("[bench] could not write %s: %s" % (t, e), file=sys.stderr)


def rccl_world1_smoke(timeout=240):
    """N = 1 only, after everything else: `python -m deeploopcloser_amd.dist --world1-smoke` as a CHILD process (a process
    group of its own, a watchdog of its own: a hang in the library's bring-up costs this key, never the line) -- a one-rank
    "nccl" group on this GPU, MatchPipeline with its collectives forced through librccl on the second stream, every batch
    compared with the one-shot call bit for bit.  Returns the compact form for the line and the full dict."""
    import subprocess
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    try:
        res = subprocess.run([sys.executable, "-m", "deeploopcloser_amd.dist", "--world1-smoke"], capture_output=True, text=True,
                             timeout=timeout, cwd=ROOT, env=env)
        rows = [l for l in res.stdout.splitlines() if l.startswith("{")]
        if res.returncode != 0 or not rows:
            return {"ok": False, "error": ("exit %d: " % res.returncode) + (res.stderr.strip().splitlines() or ["no output"])[-1][:200]}, None
        full = json.loads(rows[-1])
    except Exception as e:                                      # (a time-out included)
        return {"ok": False, "error": ("%s: %s" % (type(e).__name__, e))[:200]}, None
    cu = full.get("pipeline_collective_us") or {}
    short = {"ok": full["ok"], "backend": full["backend"], "rccl": full.get("rccl_version"), "world": full["world"],
             "init_s": _r(full["init_s"], 3), "first_collective_s": _r(full["first_collective_s"], 3),
             "allgather_us": [_r(cu.get("group_maxima"), 3), _r(cu.get("packed_topk"), 3)],
             "equals_one_shot": full["pipeline_equals_one_shot"], "exhaustive_round_equals_one_shot": full["crowded_equals_one_shot"]}
    return short, full


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: start the N rank processes (torch.distributed.run, one per
    GPU, rendezvous on 127.0.0.1) as CHILDREN of this process, which has not touched the GPU and never will; rank 0
    prints the JSON line on the inherited stdout; the children's exit code is ours."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, args.gpus))))
    sys.exit(subprocess.call(cmd, env=env))


def rccl_smoke(eng, dlc, db, pipe, queries, k, lo, world):
    """Before anything is timed: the collectives come up (a 4-byte all-reduce), and one MatchPipeline batch -- group
    maxima exchange, filtered fp64 re-score, packed all-gather, certifying merge, on the second stream -- must equal,
    bit for bit, an INDEPENDENT exchange of the same batch: every rank's own exact top-k (dlc_cosine_topk on its
    shard), two plain all-gathers, dlc_topk_merge.  Returns what goes into the bench line."""
    t0 = time.perf_counter()
    one = torch.ones(1, dtype=torch.int32, device=eng.device)
    dist.all_reduce(one)
    torch.cuda.synchronize()
    if int(one.item()) != world:
        raise SystemExit("RCCL smoke: all-reduce of ones over %d ranks gave %d" % (world, int(one.item())))
    t_up = time.perf_counter() - t0
    pipe.time_collectives = True
    for _ in range(3):
        s_p, i_p = pipe.result(pipe.submit(queries))
    coll_us = pipe.collective_us()
    pipe.time_collectives = False
    loc = eng.match_topk(db.prepare_queries(queries), db.rows, k, row_offset=lo, details=True)
    nq = loc.idx.shape[0]
    g_s = torch.empty((world, nq, k), dtype=torch.float64, device=eng.device)
    g_i = torch.empty((world, nq, k), dtype=torch.int64, device=eng.device)
    dist.all_gather_into_tensor(g_s.view(-1, k), loc.scores_f64.contiguous())
    dist.all_gather_into_tensor(g_i.view(-1, k), loc.idx.contiguous())
    m_s, m_i = eng.topk_merge(g_s, g_i)
    torch.cuda.synchronize()
    ok = bool(torch.equal(m_i, i_p) and torch.equal(m_s, s_p))
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=eng.device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) != 1:
        raise SystemExit("RCCL smoke: the pipelined sharded match differs from the plain all-gather + merge of per-shard top-k")
    try:
        rccl_version = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:                                           # (a gloo rehearsal on a box without the library)
        rccl_version = None
    if dist.get_rank() == 0:
        print("[bench] collectives up: backend %s, RCCL %s, %d ranks, first all-reduce %.2f s, all-gather #1 / #2: %s us"
              % (dist.get_backend(), rccl_version, world, t_up, coll_us), file=sys.stderr, flush=True)
    return {"backend": dist.get_backend(), "rccl_version": rccl_version, "ranks": world, "first_collective_s": t_up,
            "pipeline_equals_plain_exchange": True, "collective_us": coll_us,
            "resolved_batches": pipe.resolved_batches}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)                                    # does not return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # first contact must not hang the job: the rendezvous and every collective time out after 120 s, the failing rank
        # exits non-zero and torch.distributed.run takes the others down with it
        import datetime
        tmo = datetime.timedelta(seconds=120)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=tmo)
        else:
            dist.init_process_group("gloo", timeout=tmo)

    import deeploopcloser_amd as dlc
    from deeploopcloser_amd.engine import torch_dtype
    eng = dlc.default_engine(local_rank)
    dt = torch_dtype(args.dtype)
    n, d, nq, k = args.rows, args.dim, args.queries, args.k
    lo, hi = dlc.shard_bounds(n, world, rank)

    # ---- data: resident in HBM before anything is timed --------------------------------
    prng = np.random.RandomState(4321)
    planted_rows = prng.choice(n, nq, replace=False)
    rows, planted = synth_shard(eng, n, d, lo, hi, dt, planted_rows)
    g = torch.Generator(device=eng.device)
    g.manual_seed(99)
    noise = torch.randn((nq, d), generator=g, device=eng.device)
    # planted neighbour at cosine ~0.9: centred U(0,1) rows have norm sqrt(d/12); sigma from that
    sigma = float(np.sqrt(1.0 / 12.0) * np.sqrt(1 / 0.81 - 1))
    queries = eng.normalize(planted + sigma * noise, dt, center=True)
    crowded_rows = None
    if args.crowded:
        # kg * 8 + 1 exact copies of query 0's planted row, evenly spread: the kg best groups hold at most kg * 8 of them, one
        # more is always left behind with the SAME score as the k-th -- the certificate must refuse, on every rank count
        copies = eng.groups_per_query(k) * 8 + 1
        crowded_rows = np.unique(np.linspace(0, n - 1, copies).astype(np.int64))
        src = eng.normalize(planted[0:1], dt, center=True)                 # the stored bits of that row, the same on every rank
        mine = torch.as_tensor(crowded_rows[(crowded_rows >= lo) & (crowded_rows < hi)] - lo, device=eng.device)
        rows[mine] = src
    db = dlc.KeyframeDatabase(rows, dtype=dt, row_offset=lo, stored=True)
    sharded = dlc.ShardedKeyframeDatabase.from_database(db)
    use_pipe = not args.no_pipeline and (world > 1 or args.pipeline)
    pipe = dlc.MatchPipeline(db, k, depth=2 if world == 1 else 3) if use_pipe else None
    torch.cuda.synchronize()
    smoke = None
    if world > 1:
        smoke = rccl_smoke(eng, dlc, db, pipe if pipe is not None else dlc.MatchPipeline(db, k, depth=1), queries, k, lo, world)
    last = [None]

    def step():
        # one query batch against the whole (sharded) database.  Pipelined mode: the batch's
        # GEMM is enqueued now; its selection / all-gather / merge run on the second stream
        # and complete before the closing fence (torch.cuda.synchronize waits for all streams).
        if pipe is None:
            return sharded.match_topk(queries, k)
        t = last[0] = pipe.submit(queries)
        # every batch's result is fetched, depth - 1 batches behind the submission (the pipeline stays full): an
        # uncertified batch's exhaustive round (collectives + fp64 pass, MatchPipeline._resolve) runs inside result()
        # and so inside the timed region, and resolved_batches counts them all
        if t >= pipe.depth - 1:
            pipe.result(t - (pipe.depth - 1))
        return None

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    import gc
    gc.collect()                                             # no cyclic collection inside the timed region -- and none between
    gc.disable()                                             # the warm-up and it: a collection here is tens of milliseconds of an
    eng.set_profiling(True)                                  # idle GPU, and the first steps behind an idle stretch run at the
    try:                                                     # clock the chip dropped to (docs/LAB.md 11.6)
        for _ in range(args.warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = step()
        fence()
        t1 = time.perf_counter()
    finally:
        gc.enable()
    scores, idx = res if pipe is None else pipe.result(last[0])
    gemm_ms = eng.profile_gemm_ms(min(args.steps, 256))
    eng.set_profiling(False)

    # ---- the step's second stage on its own (N=1, untimed above): selection + fp64 re-score + certificate (+ the early-exit
    # launch of the exhaustive pass), HIP events around dlc_cosine_select_topk behind a score pass into the same workspace
    finish_ms = None
    if world == 1 and pipe is None:
        qs = db.prepare_queries(queries)
        ws2 = torch.empty(eng.topk_workspace_bytes(nq, db.rows.shape[0], d, k), dtype=torch.uint8, device=eng.device)
        o_s = torch.empty((nq, k), dtype=torch.float32, device=eng.device)
        o_i = torch.empty((nq, k), dtype=torch.int64, device=eng.device)
        tms = []
        for _ in range(12):
            eng.score_groups(qs, db.rows, k, ws2)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            eng.select_topk(qs, db.rows, k, ws2, o_s, o_i, row_offset=lo)
            e1.record()
            torch.cuda.synchronize()
            tms.append(e0.elapsed_time(e1))
        finish_ms = float(np.mean(tms[2:]))
        del ws2, o_s, o_i

    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=eng.device)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    ms_per_step = elapsed / args.steps * 1e3
    qps = nq * args.steps / elapsed

    # ---- quality: recall@1 on the planted neighbours -------------------------------------
    recall1 = float((idx[:, 0].cpu().numpy() == planted_rows).mean())
    if crowded_rows is not None:                                # query 0's neighbour has copies: any of them is the right answer
        hit = idx[:, 0].cpu().numpy() == planted_rows
        hit[0] = bool(np.isin(idx[0, 0].item(), crowded_rows)) or hit[0]
        recall1 = float(hit.mean())
    import hashlib
    idx_sha = hashlib.sha256(np.ascontiguousarray(idx.cpu().numpy()).tobytes()).hexdigest()
    scores_sha = hashlib.sha256(np.ascontiguousarray(scores.cpu().numpy()).tobytes()).hexdigest()

    out = None
    if rank == 0:
        e = 2
        gemm_avg_ms = float(np.mean(gemm_ms)) if gemm_ms else float("nan")
        shard_rows = hi - lo
        algo_bytes = shard_rows * d * e + nq * d * e            # DB shard read once + the query block
        flops = 2.0 * nq * shard_rows * d
        achieved_gbs = algo_bytes / (gemm_avg_ms * 1e-3) / 1e9
        achieved_tf = flops / (gemm_avg_ms * 1e-3) / 1e12
        traffic = pmc_traffic(n, d, nq, args.dtype, world)
        out = {
            "metric": "query-frames/sec vs N-keyframe DB + top-k recall@1",
            "value": qps, "unit": "query-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%d-keyframe synthetic %d-d descriptor DB (%s, L2-normalised), batch=%d queries, "
                                   "top-%d cosine match; DB row-sharded over %d GPU(s), RCCL all-gather of per-shard "
                                   "top-k (BASELINE configs[4] shape, bf16 per north_star)" % (n, d, args.dtype, nq, k, world),
                       "db_rows": n, "dim": d, "queries_per_step": nq, "k": k, "rows_per_gpu": shard_rows,
                       "pipelined": pipe is not None, "crowded": bool(args.crowded)},
            "recall_at_1": recall1,
            # the step = score GEMM (roofline.kernel_ms) + this: selection, fp64 re-score of the candidates, certificate
            "finish_ms": finish_ms, "step_minus_gemm_ms": ms_per_step - (float(np.mean(gemm_ms)) if gemm_ms else float("nan")),
            # digests of the timed result (the same database and queries whatever --gpus is): equal across rank counts
            "topk_idx_sha256": idx_sha, "topk_scores_sha256": scores_sha,
            "rccl_ranks": world if world > 1 else None, "rccl_smoke": smoke,
            "pipeline": {"depth": pipe.depth, "resolved_batches": pipe.resolved_batches,
                         "dropped_batches": pipe.dropped_batches} if pipe is not None else None,
            "collective_us": smoke["collective_us"] if smoke else None,
            # traffic: HBM bytes per launch from the rocprofv3 PMC passes of this same command, as committed under
            # profiles/ (bench.py cannot run the profiler on itself): a REPLAYED figure, not measured in this run
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS,
                         # ... and against what a device copy reaches on this part (6.29 TB/s): the power-capped operating point
                         # read against an achievable stream rather than the pin rate
                         "frac_of_measured_copy": achieved_gbs / HBM_COPY_GBS, "measured_copy_gbs": HBM_COPY_GBS,
                         "traffic": traffic[0] if traffic else None,
                         "traffic_profiled": traffic[0] if traffic else None,
                         "traffic_source": ("profiles/%s (separate rocprofv3 --pmc passes of this command, replayed here -- "
                                            "not measured in this run)" % traffic[1]) if traffic else None,
                         "kernel": "score_gemm_kernel", "kernel_ms": gemm_avg_ms, "kernel_launches_timed": len(gemm_ms),
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "mfma_achieved_tflops": achieved_tf, "mfma_peak_tflops": MFMA_PEAK_TFLOPS,
                         "mfma_frac": achieved_tf / MFMA_PEAK_TFLOPS},
        }

    # ---- package power / clock while the same loop runs on (untimed; rank 0, N=1 only): the score GEMM
    # sits on the board's power cap, which is what bounds it -------------------------
    if rank == 0 and world == 1 and not args.no_power_probe:
        pp = out["roofline"]["power_probe"] = power_probe(step, seconds=2.5)
        # joules per step = the probe's median package power x the probe loop's own mean step (the same loop, the same
        # operating point); the board's energy accumulator over the same stretch rides along in power_probe when it exists
        out["roofline"]["energy_j_per_step"] = pp["package_power_w"] * pp["ms_per_step"] * 1e-3 if pp else None
        # ... and the same K steps timed once more, now that the chip has been under THIS load for 2.5 s: after any pause the
        # first ~25 launches of the score GEMM run ~5 % slower (2.05 ms against 1.96: scripts/exp_step_series.py), which is
        # where a `--warmup 5 --steps 20` run has its whole timed region.  `value` above stays the contract's number; this is
        # what a resident service sees.
        fence()
        t0s = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dts = time.perf_counter() - t0s
        out["steady_state"] = {"value": nq * args.steps / dts, "unit": "query-frames/s", "ms_per_step": dts / args.steps * 1e3,
                               "steps": args.steps, "measured": "the same loop again, behind the power probe's 2.5 s of the same steps"}

    # ---- what ONE rank of a 2 / 4 / 8-GPU run would do per batch, emulated on this GPU (no RCCL): driver-timed every
    # round, since an 8-GPU node is not always at hand (rank 0, N=1 only; untimed above) ------------------------------
    if rank == 0 and world == 1 and not args.no_shard_emulation and n >= 8 * 2048:
        out["multi_gpu_emulation"] = bench_shard_emulation(eng, dlc, db.rows, queries, k, idx)

    # ---- BASELINE configs[3] and configs[4] on this one GPU, each a timed row of its own, and one rank's step of their
    # 8-GPU form (rank 0, N=1 only; untimed above) -------------------------------------------------------------------
    if rank == 0 and world == 1 and not args.no_configs and not args.crowded:
        out["baseline_configs"] = bench_configs(eng, dlc, args, db.rows, queries, planted, noise, sigma, planted_rows)

    # ---- the other rows of the hot path at configs[1] / configs[2] size (rank 0, N=1 only; untimed above) ----
    if rank == 0 and world == 1 and not args.no_paths:
        del rows, planted, noise
        out["paths"] = bench_paths(eng, args.path_frames)

    # ---- CPU baseline + index agreement on a bounded sample (rank 0, N=1 only).  LAST: its BLAS / torch thread pools keep
    # spinning for a while and took 10 ms out of the host-to-host rows above when it ran in front of them ----------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import cosine as ocos
        ns = min(args.cpu_sample_rows, n)
        qh = queries.float().cpu().numpy().astype(np.float64)
        best_s = np.empty((nq, 0))
        best_i = np.empty((nq, 0), dtype=np.int64)
        t_cpu = 0.0
        blk = 65536
        for b0 in range(0, ns, blk):                        # the host copy / upcast of a block is not timed
            dbh = db.rows[b0:min(b0 + blk, ns)].float().cpu().numpy().astype(np.float64)
            t0 = time.perf_counter()
            bs, bi = ocos.cosine_topk(qh, dbh, k, row_offset=b0)
            best_s, best_i = ocos.merge_topk(np.concatenate([best_s, bs], 1), np.concatenate([best_i, bi], 1), k)
            t_cpu += time.perf_counter() - t0
            del dbh
        if ns == n:
            s_gpu, i_gpu = scores, idx                      # the timed result itself
        else:
            s_gpu, i_gpu = eng.match_topk(queries, db.rows[:ns], k)
        torch.cuda.synchronize()
        agree = float((i_gpu.cpu().numpy() == best_i).mean())
        # the same oracle in fp32 (BASELINE.md section 4(1)'s plan) on the first blocks only, scaled linearly in rows
        ns32 = min(ns, 4 * blk)
        q32 = qh.astype(np.float32)
        t32 = 0.0
        for b0 in range(0, ns32, blk):
            dbh = db.rows[b0:min(b0 + blk, ns32)].float().cpu().numpy()
            t0 = time.perf_counter()
            ocos.cosine_topk(q32, dbh, k, row_offset=b0, dtype=np.float32)
            t32 += time.perf_counter() - t0
            del dbh
        # ... and BASELINE.md section 4(1)'s other form: torch-CPU fp32 matmul + topk over ALL host cores (NumPy's BLAS pool
        # stops at 64 threads), same first blocks, scaled linearly in rows
        ncpu = os.cpu_count() or 1
        prev_threads = torch.get_num_threads()
        torch.set_num_threads(ncpu)
        qt = torch.from_numpy(q32)
        t_torch = 0.0
        for rep in range(2):                                 # (the first pass warms the thread pool)
            t_torch = 0.0
            for b0 in range(0, ns32, blk):
                dbt = db.rows[b0:min(b0 + blk, ns32)].float().cpu()
                t0 = time.perf_counter()
                torch.topk(qt @ dbt.T, min(k, dbt.shape[0]), dim=1)
                t_torch += time.perf_counter() - t0
                del dbt
        torch.set_num_threads(prev_threads)
        out["cpu_baseline"] = {
            "value": nq / (t_cpu * (n / ns)), "unit": "query-frames/s", "cores": blas_threads(),
            "value_torch_f32_all_cores": nq / (t_torch * (n / ns32)), "torch_threads": ncpu,
            "host_cpus": os.cpu_count(), "kind": "port", "arithmetic": "f64",
            "value_f32": nq / (t32 * (n / ns32)),
            "sample_short": "oracle/cosine.py fp64 NumPy matmul + exact top-%d, %d queries x %d of %d rows: %.1f s CPU%s"
                            % (k, nq, ns, n, t_cpu, "" if ns == n else ", scaled linearly in rows"),
            "sample": "oracle/cosine.py (NumPy matmul on the BLAS pool of `cores` threads + exact top-k, blocks of %d rows) "
                      "on %d queries x %d of %d DB rows in fp64: %.1f s of CPU work%s; value_f32: the same in fp32 on the "
                      "first %d rows (%.1f s), scaled linearly in DB rows; value_torch_f32_all_cores: torch-CPU fp32 matmul + topk on "
                      "%d threads over the same first rows (%.2f s), scaled the same way"
                      % (blk, nq, ns, n, t_cpu, "" if ns == n else "; scaled linearly in DB rows", ns32, t32, ncpu, t_torch)}
        out["topk_index_agreement_vs_oracle"] = agree
        out["topk_index_agreement_rows"] = ns
        out["topk_score_max_abs_err_vs_oracle"] = float(np.abs(s_gpu.cpu().numpy() - best_s).max())

    if rank == 0 and world == 1 and not args.no_rccl_smoke:
        out["rccl_world1_smoke"], out["rccl_world1_smoke_full"] = rccl_world1_smoke()

    # the collectives come down FIRST (whatever the library prints on the way out must not follow the line), then rank 0
    # writes the sidecar file and prints the ONE line: the last thing on stdout
    try:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
    finally:
        if rank == 0:
            write_detail(out, args.detail)
            sys.stdout.flush()
            sys.stderr.flush()
            print(compact_line(out, os.path.basename(args.detail) if args.detail else DETAIL_FILE), flush=True)


if __name__ == "__main__":
    main()
