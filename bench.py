#!/usr/bin/env python3
"""Bench of the trie-constrained beam-search retrieval path on MI355X.

A "step" is one pass of the hot path (T5 encoder + L fused decode/select steps, rpr_search) over
one batch of Q synthetic MSMARCO-dev-shaped queries; the workload is BASELINE.json configs[1]:
t5-base dims, 8 841 823-doc synthetic docid trie (32 x 256 codes), beam = 10, len = 32, fp32.
Inputs (token ids / masks of every step's batch, weights, trie) are resident in HBM before the
timed region. value = queries/s over all ranks (weak scaling: every rank runs K steps of Q queries
on its own shard, model + trie replicated, one RCCL all_gather of the ranked results at the end).

  python bench.py [--gpus N --steps K --warmup W --batch Q --beams B --len L --docs N_DOCS]

Extra legs on rank 0 (outside the timed region):
  roofline     one eager step of the TIMED configuration with hipEvents around every launch on the stream it is
               launched on (the library's profile mode; nothing synchronises while the step is enqueued, so the two
               lanes run side by side as in the timed region); the dominant kernel is the split-precision 256x256
               ping-pong GEMM (gemm_h2_pp_kernel). The rocprofv3 average of the same kernel in the same configuration
               is read from profiles/latest_kernel_stats.csv and reported beside it (roofline.source_profile).
  cpu_baseline the oracle "port" of the reference loop (no KV cache, dict+CSR float64 mask, top-2B,
               Python scorer) on the host cores, on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

MSMARCO_DOCS = 8_841_823
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2500.0  # dense f16/bf16 MFMA peak (2:1-sparsity figures are never used)
PEAK_HBM_TBS = 8.0


def _build_id():
    """sha256 prefix of the sources libripor_hip.so was built from (the stamp __graft_entry__.build() writes beside the library)."""
    try:
        return open(os.path.join(REPO, "ripor_amd", "libripor_hip.so.srchash")).read().strip()[:16]
    except OSError:
        return "unknown"


BUILD_ID = _build_id()


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def algorithmic_bytes_per_query(dims, Q, B, L, Lq, s_w=4, s_kv=4):
    """SURVEY.md §8(d) formula (KV-cached algorithm, weights read once per step per batch of Q)."""
    d, inner, dff = dims.d_model, dims.inner, dims.d_ff
    ne, nd, V = dims.num_layers, dims.num_decoder_layers, dims.decoder_vocab_sizes[0]
    w_enc = ne * (4 * d * inner + 2 * d * dff)
    w_xkv = nd * 2 * d * inner
    w_dec = nd * (6 * d * inner + 2 * d * dff)
    kvrow = nd * 2 * inner
    weights = ((w_enc + w_xkv) * s_w + L * w_dec * s_w + L * V * d * s_w) / Q
    self_r = B * (L * (L + 1) / 2) * kvrow * s_kv
    self_w = B * L * kvrow * s_kv
    cross_r = L * Lq * kvrow * s_kv
    cross_w = Lq * kvrow * s_kv
    trie = B * L * 40
    out = B * (4 * L + 4)
    return weights + self_r + self_w + cross_r + cross_w + trie + out


def algorithmic_forced_tail_per_query(dims, Q, B, L, Lq, T, s_w=4, s_act=4):
    """The same accounting for the forced-tail algorithm with its (first) fork at depth T: T sequential KV-cached steps as in
    §8(d), then ONE teacher-forced pass over the B * (L - T) remaining positions — decoder weights read once more per batch,
    every tail row's q/k/v and attention output written once and read once, the query's cross K/V read once per layer for
    the whole pass, one gold logit per position. Returns (bytes, flops)."""
    d, inner, dff = dims.d_model, dims.inner, dims.d_ff
    ne, nd, V = dims.num_layers, dims.num_decoder_layers, dims.decoder_vocab_sizes[0]
    T = max(0, min(int(T), L))
    w_enc = ne * (4 * d * inner + 2 * d * dff)
    w_xkv = nd * 2 * d * inner
    w_dec = nd * (6 * d * inner + 2 * d * dff)
    kvrow = nd * 2 * inner
    rows = B * (L - T)
    weights = ((w_enc + w_xkv) * s_w + (T + (1 if rows else 0)) * w_dec * s_w + T * V * d * s_w) / Q
    steps_kv = B * (T * (T + 1) / 2) * kvrow * s_act + B * T * kvrow * s_act + T * Lq * kvrow * s_act + Lq * kvrow * s_act
    tail_act = rows * nd * (3 * inner + inner + inner + inner) * s_act * 2     # qkv, self out, cross q, cross out: write + read
    tail_prefix_kv = B * T * kvrow * s_act                                      # the T cached positions, once per beam
    tail_cross = (Lq * kvrow * s_act) if rows else 0
    gold = rows * d * s_w                                                        # one codebook row per position
    byts = weights + steps_kv + tail_act + tail_prefix_kv + tail_cross + gold + B * T * 40 + B * (4 * L + 4)
    attn_self = 4 * inner * nd * B * (T * (T + 1) / 2 + (L - T) * T + (L - T) * (L - T + 1) / 2)
    flops = (2 * w_enc * Lq + 2 * w_xkv * Lq + B * L * 2 * w_dec + B * T * 2 * V * d + rows * 2 * d
             + 4 * Lq * Lq * inner * ne + attn_self + 4 * inner * nd * B * L * Lq)
    return byts, flops


def algorithmic_flops_per_query(dims, B, L, Lq):
    d, inner, dff = dims.d_model, dims.inner, dims.d_ff
    ne, nd, V = dims.num_layers, dims.num_decoder_layers, dims.decoder_vocab_sizes[0]
    w_enc = ne * (4 * d * inner + 2 * d * dff)
    w_xkv = nd * 2 * d * inner
    w_dec = nd * (6 * d * inner + 2 * d * dff)
    return (2 * w_enc * Lq + 2 * w_xkv * Lq + B * L * 2 * w_dec + B * L * 2 * V * d
            + 4 * Lq * Lq * inner * ne + 4 * inner * nd * B * L * (L + 1) / 2 + 4 * inner * nd * B * L * Lq)


def cpu_baseline(sd, dims, B, L, n_queries=16, trie_docs=10_000):
    """Reference-faithful CPU loop (oracle 'port') on the host cores; bounded sample.

    The GPU box has far more cores than these small fp32 GEMMs can use (torch CPU gets *slower*
    beyond a few dozen threads), so the thread count is chosen by timing a short warm-up at
    8/16/32/64 threads and keeping the fastest; ``cores`` reports the threads actually used."""
    from oracle import beam_ref, t5_ref
    from ripor_amd.utils import synth
    host_cores = len(os.sched_getaffinity(0))
    V = dims.decoder_vocab_sizes[0]
    codes = synth.make_codes(trie_docs, L, V)
    pm = beam_ref.PrefixMaskRef(beam_ref.build_list_smtid_to_nextids(synth.codes_to_docid_to_smtid(codes)), V)
    model = t5_ref.T5Ref(sd, dims)
    ids, mask = synth.make_queries(n_queries + 1, vocab_size=dims.vocab_size, seed=77)
    best_t, best_dt, probe = None, None, {}
    for th in [t for t in (8, 16, 32, 64) if t <= host_cores] or [host_cores]:
        torch.set_num_threads(th)
        beam_ref.beam_search_ref(model, pm, ids[:1], mask[:1], B, 2)  # thread-pool spin-up
        t0 = time.time()
        beam_ref.beam_search_ref(model, pm, ids[:1], mask[:1], B, min(L, 6))
        dt = time.time() - t0
        log(f"[bench] cpu_baseline probe: {th} threads -> {dt:.2f}s (1 query, {min(L, 6)} steps)")
        probe[str(th)] = round(dt, 3)
        if best_dt is None or dt < best_dt:
            best_t, best_dt = th, dt
    torch.set_num_threads(best_t)
    t0 = time.time()
    beam_ref.beam_search_ref(model, pm, ids[1:], mask[1:], B, L)
    dt = time.time() - t0
    log(f"[bench] cpu_baseline: {n_queries} queries in {dt:.1f}s on {best_t} threads")
    # the reference retrieval script runs batch_size=1 (full_evaluate_t5seq_aq_encoder.sh:197): three single-query calls
    t1 = time.time()
    for i in range(3):
        beam_ref.beam_search_ref(model, pm, ids[1 + i:2 + i], mask[1 + i:2 + i], B, L)
    dt1 = (time.time() - t1) / 3
    log(f"[bench] cpu_baseline at batch 1: {dt1:.2f}s per query")
    return {"value": n_queries / dt, "unit": "queries/s", "cores": best_t, "kind": "port",
            "value_batch1": 1.0 / dt1, "host_cores": host_cores,
            "thread_probe_s": probe,   # seconds for 1 query x min(L, 6) steps at each thread count tried; `cores` = the fastest

            "sample": f"{n_queries} queries in one batch, t5-base dims fp32, beams={B}, len={L}, "
                      f"{trie_docs}-doc dict+CSR trie (the reference's dict structure for 8.8M docs does not fit "
                      f"host RAM), full-prefix decoder recompute like the reference (no KV cache); "
                      f"{dt:.1f}s wall on {best_t} threads (fastest of 8/16/32/64; host has {host_cores} cores); "
                      f"value_batch1 = the same loop at the reference script's batch size 1 ({dt1:.2f}s per query)"}



# --------------------------------------------------------------------------------------------------------------------
# Secondary legs (outside the timed region, a few seconds each): the other BASELINE configurations and SURVEY §8 "next"
# rows, so that the driver's bench record carries a number for each of them. Same library, same C ABI.

def _time_search(E, model, trie, batches, B, L, steps, warmup=1):
    for i in range(warmup):
        E.search(model, trie, batches[i % len(batches)][0], batches[i % len(batches)][1], B, L)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        r = E.search(model, trie, batches[(warmup + i) % len(batches)][0], batches[(warmup + i) % len(batches)][1], B, L)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, r


def _query_batches(synth, dims, Q, n, dev, seed):
    ids, mask = synth.make_queries(Q * n, vocab_size=dims.vocab_size, seed=seed)
    lq = (int(mask.sum(1).max()) + 7) // 8 * 8
    ids = np.pad(ids, ((0, 0), (0, max(0, lq - ids.shape[1]))))[:, :lq]
    mask = np.pad(mask, ((0, 0), (0, max(0, lq - mask.shape[1]))))[:, :lq]
    return [(torch.from_numpy(ids[i * Q:(i + 1) * Q]).to(dev, torch.int32), torch.from_numpy(mask[i * Q:(i + 1) * Q]).to(dev, torch.int32))
            for i in range(n)]


def _leftover_guard(ctx, fn):
    """Run fn() in the ctx's forced-tail mode; if the optimistic mode left a query unforced, repeat in the exact mode."""
    ctx.status(clear=True)
    out = fn()
    if ctx.status(clear=True) & 4:
        mode = ctx.forced_tail()
        ctx.set_forced_tail(1)
        try:
            out = fn()
        finally:
            ctx.set_forced_tail(mode)
        out = (out, True)
    else:
        out = (out, False)
    return out


def _profiled(ctx, fn):
    """One eager pass of fn() with hipEvents around every launch (the library's profile mode) -> per-class stats."""
    ctx.profile_reset()
    ctx.profile_enable(True)
    try:
        fn()
        torch.cuda.synchronize()
        return ctx.profile_get()
    finally:
        ctx.profile_enable(False)


def _gemm_roofline(stats, peak_tflops, cu_fraction, kernel, note):
    """`roofline` object of a secondary leg: the GEMM launches of one profiled pass (dominant + small-tile kernels together:
    these legs run mixed tile shapes) against the MFMA peak of the arithmetic they use."""
    g, gs = stats["gemm"], stats["gemm_small"]
    ms, fl, n = g["total_ms"] + gs["total_ms"], g["flops"] + gs["flops"], g["launches"] + gs["launches"]
    tot = sum(v["total_ms"] for v in stats.values())
    ach = fl / max(1e-9, ms * 1e-3) / 1e12
    return {"kernel": kernel, "bound": "mfma", "achieved": ach, "peak": peak_tflops * cu_fraction, "unit": "TFLOP/s",
            "frac": ach / (peak_tflops * cu_fraction), "cu_fraction": cu_fraction, "traffic": None,
            "gemm_launches": n, "avg_launch_us": ms * 1e3 / max(1, n), "gemm_ms": ms, "all_kernels_ms": tot,
            "gemm_share_of_kernel_time": ms / max(1e-9, tot), "note": note}


def secondary_config4(E, synth, ctx, trie, dev, L, queries=162, steps=5):
    """BASELINE config 4: t5-large dims, the same 8.8M-doc trie, beam 100, len 32."""
    dims = synth.t5_large_dims(L=L)
    t0 = time.time()
    model = E.DeviceModel(ctx, synth.make_state_dict(dims), dims)
    t_w = time.time() - t0
    batches = _query_batches(synth, dims, queries, 2, dev, seed=404)
    (dt, r), fb = _leftover_guard(ctx, lambda: _time_search(E, model, trie, batches, 100, L, steps))
    ok = int((r.row_hi > r.row_lo).sum().item())
    lanes = 0 < ctx.lane_split() <= queries * 100
    st = _profiled(ctx, lambda: E.search(model, trie, batches[0][0], batches[0][1], 100, L))
    roof = _gemm_roofline(st, PEAK_F16_MFMA_TFLOPS / 3.0, 0.5 if lanes else 1.0, "rpr::gemm_h2_pp_kernel (+ small-tile kernels)",
                          "algorithmic 2MNK flops of every projection launch of one step / their summed event durations; 3 f16 "
                          "MFMAs per product -> peak 2500/3 TF/s, halved per launch when the step runs as two CU-masked lanes")
    out = {"workload": f"t5-large dims (d 1024, d_ff 4096, 24+24 layers), {trie.N}-doc trie, beams=100, len={L}, {queries} queries/step",
           "value": queries / dt, "unit": "queries/s", "ms_per_step": dt * 1e3, "steps": steps, "queries_per_step": queries,
           "roofline": roof, "lanes": 2 if lanes else 1,
           "dtype": "f32 via f16x2-split MFMA (fp32 accumulate)", "valid_leaves": f"{ok}/{queries * 100}",
           "forks_last_step": ctx.last_fork_stats(), "leftover_fallback_taken": fb, "weights_s": round(t_w, 1)}
    del model
    return out


def secondary_t5_3b(E, synth, ctx, trie, dev, L, queries=384, steps=2):
    """t5-3b dims (the third model size the reference's constructor accepts, t5_generative_retriever.py:128-133: d 1024, 32 heads of
    d_kv 128, d_ff 16384, 24 + 24 layers), the same trie, beam 10: the 128-dim head path (generic attention kernels, no forced
    tail). Opt-in (--secondary ...,t5_3b): 23 GB of weights + planes and 250 MB of self-attention K/V per query in flight."""
    dims = synth.ModelDims(d_model=1024, d_kv=128, d_ff=16384, num_layers=24, num_decoder_layers=24, num_heads=32,
                           decoder_vocab_sizes=[256] * L)
    t0 = time.time()
    model = E.DeviceModel(ctx, synth.make_state_dict(dims), dims)
    t_w = time.time() - t0
    batches = _query_batches(synth, dims, queries, 2, dev, seed=405)
    dt, r = _time_search(E, model, trie, batches, 10, L, steps)
    ok = int((r.row_hi > r.row_lo).sum().item())
    st = _profiled(ctx, lambda: E.search(model, trie, batches[0][0], batches[0][1], 10, L))
    lanes = 0 < ctx.lane_split() <= queries * 10
    roof = _gemm_roofline(st, PEAK_F16_MFMA_TFLOPS / 3.0, 0.5 if lanes else 1.0, "rpr::gemm_h2_pp_kernel (+ small-tile kernels)",
                          "algorithmic 2MNK flops of every projection launch of one step / their summed event durations; 3 f16 "
                          "MFMAs per product -> peak 2500/3 TF/s, halved per launch when the step runs as two CU-masked lanes")
    by_class = {k: round(v["total_ms"], 2) for k, v in st.items() if v["launches"]}
    out = {"workload": f"t5-3b dims (d 1024, 32 heads x 128, d_ff 16384, 24+24 layers), {trie.N}-doc trie, beams=10, len={L}, {queries} queries/step",
           "value": queries / dt, "unit": "queries/s", "ms_per_step": dt * 1e3, "steps": steps, "queries_per_step": queries,
           "roofline": roof, "lanes": 2 if lanes else 1, "kernel_ms_by_class": by_class,
           "dtype": "f32 via f16x2-split MFMA (fp32 accumulate)", "valid_leaves": f"{ok}/{queries * 10}", "weights_s": round(t_w, 1)}
    del model
    return out


def board_power_under(run, seconds=2.5):
    """Board power and shader clock while `run()` keeps the GPU busy (outside every timed region): `rocm-smi --showpower
    --showclocks --showuse` sampled from a thread. Best effort — None when rocm-smi is missing or prints something else."""
    import re, shutil, subprocess, threading
    if not shutil.which("rocm-smi"):
        return None
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            try:
                txt = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showuse"], capture_output=True, text=True, timeout=5).stdout
                pw = re.search(r"Power \(W\): ([0-9.]+)", txt)
                sc = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", txt)
                use = re.search(r"GPU use \(%\): (\d+)", txt)
                if pw and sc and use:
                    samples.append((float(pw.group(1)), int(sc.group(1)), int(use.group(1))))
            except Exception:
                return

    cap = None
    try:
        m = re.search(r"Max Graphics Package Power \(W\): ([0-9.]+)",
                      subprocess.run(["rocm-smi", "--showmaxpower"], capture_output=True, text=True, timeout=5).stdout)
        cap = float(m.group(1)) if m else None
    except Exception:
        pass
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    t0 = time.time()
    while time.time() - t0 < seconds:
        run()
        torch.cuda.synchronize()
    stop.set()
    th.join(timeout=6)
    busy = [x for x in samples if x[2] >= 90]
    if not busy:
        return None
    return {"samples_busy": len(busy), "power_w_mean": sum(b[0] for b in busy) / len(busy), "power_w_max": max(b[0] for b in busy),
            "power_cap_w": cap, "sclk_mhz_mean": sum(b[1] for b in busy) / len(busy), "sclk_mhz_min": min(b[1] for b in busy),
            "sclk_mhz_nominal": 2400,
            "note": "rocm-smi sampled while the headline step repeats (outside the timed region; GPU use >= 90 % samples only): "
                    "power_w_mean against power_cap_w and sclk_mhz_mean against the 2400 MHz nominal clock say how far DVFS throttles "
                    "the GEMM-dominated step (default batch: ~1340 W of 1400 W at ~1.85 GHz, profiles/archive/r03e_power_samples.txt)"}


def secondary_latency(E, synth, ctx, model, trie, dims, dev, L, steps=8):
    """One query in flight: the headline beam (10) and the reference script's literal setting (full_evaluate_t5seq_aq_encoder.sh
    runs evaluate.py with --batch_size 1 --topk 1000): ms per query, every search a hipGraph replay."""
    out = {"workload": f"t5-base dims, {trie.N}-doc trie, len={L}, 1 query per search", "unit": "ms/query"}
    batches = _query_batches(synth, dims, 1, 4, dev, seed=303)
    for beams in (10, 1000):
        (dt, r), fb = _leftover_guard(ctx, lambda: _time_search(E, model, trie, batches, beams, L, steps, warmup=2))
        out[f"beams{beams}"] = {"value": dt * 1e3, "forks_last_step": ctx.last_fork_stats(), "leftover_fallback_taken": fb,
                                "valid_leaves": f"{int((r.row_hi > r.row_lo).sum().item())}/{beams}"}
    return out


def _small_batch_traffic(key):
    """(bytes per search, provenance) of a small-batch configuration from the committed PMC passes (profiles/latest_small_batch_pmc.json,
    written by tools/small_batch_pmc.sh: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc runs, summed over every rpr:: kernel)."""
    try:
        pmc = json.load(open(os.path.join(REPO, "profiles", "latest_small_batch_pmc.json")))
        e = pmc[key]
        b = (2.0 * e["fetch_kb_per_search"] + e["write_kb_per_search"]) * 1024.0
        return b, (f"profiles/latest_small_batch_pmc.json[{key}]: (2*FETCH_SIZE + WRITE_SIZE)*1024 over all kernels of one search; "
                   f"counters of build {pmc.get('_meta', {}).get('build')}, this run is build {BUILD_ID}")
    except Exception:
        return None, None


def secondary_small_batch(E, synth, ctx, model, trie, dims, dev, B, L, steps=8):
    """the regime north_star's "fraction of the HBM roofline" is about — 1, 8 and 64 queries in flight
    (the reference script's literal setting is batch 1). Per batch size: queries/s, ms per search, and a `roofline` object
    of bound "hbm": achieved = the bytes of the forced-tail algorithm actually run (first fork from the search itself) x
    queries/s against the 8 TB/s peak (`*_8d_equivalent`: SURVEY §8(d)'s bytes of the step-by-step algorithm — an equivalent-work
    ratio), `traffic` from the committed PMC passes, plus the GEMM launches' share of the search from one
    event-timed eager pass (`gemm_ms`, `launches`)."""
    out = {"workload": f"t5-base dims, {trie.N}-doc trie, beams={B}, len={L}, Q queries per search, hipGraph replay", "unit": "queries/s"}
    for Q in (1, 8, 64):
        batches = _query_batches(synth, dims, Q, 4, dev, seed=404 + Q)
        Lq = int(batches[0][0].shape[1])
        lq_mean = float(sum(float(b[1].sum()) for b in batches) / (len(batches) * Q))
        (dt, r), fb = _leftover_guard(ctx, lambda: _time_search(E, model, trie, batches, B, L, steps, warmup=2))
        forks = ctx.last_fork_stats()
        T = forks[0]["depth"] if forks else L
        by_8d = algorithmic_bytes_per_query(dims, Q, B, L, lq_mean)
        by_ft, fl_ft = algorithmic_forced_tail_per_query(dims, Q, B, L, lq_mean, T)
        stats = _profiled(ctx, lambda: E.search(model, trie, batches[0][0], batches[0][1], B, L, use_graph=False))
        n_launch = sum(int(v["launches"]) for v in stats.values())
        gemm_ms = stats["gemm"]["total_ms"] + stats["gemm_small"]["total_ms"]
        qps = Q / dt
        traffic, traffic_src = _small_batch_traffic(f"q{Q}_b{B}")
        out[f"q{Q}"] = {
            "value": qps, "ms_per_search": dt * 1e3, "padded_len": Lq, "mean_query_tokens": lq_mean,
            "forks_last_search": forks, "leftover_fallback_taken": fb, "launches_per_search": n_launch,
            "gemm_ms_event_timed": gemm_ms,
            "roofline": {"bound": "hbm", "achieved": by_ft * qps / 1e9, "peak": PEAK_HBM_TBS * 1e3, "unit": "GB/s",
                         "frac": by_ft * qps / 1e12 / PEAK_HBM_TBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_query": by_ft,
                         "achieved_8d_equivalent": by_8d * qps / 1e9, "frac_8d_equivalent": by_8d * qps / 1e12 / PEAK_HBM_TBS,
                         "algorithmic_bytes_per_query_8d": by_8d,
                         "flops_per_query_forced_tail": fl_ft,
                         "mfma_frac_forced_tail": fl_ft * qps / 1e12 / (PEAK_F16_MFMA_TFLOPS / 3.0),
                         "note": "whole-search figure (the path at this batch size is a chain of dependent launches, no single "
                                 "kernel dominates). achieved / frac = the bytes of the forced-tail algorithm actually run (weights "
                                 "streamed fork_depth + 1 times, fp32-sized operands) x queries/s: the bandwidth the search really "
                                 "sustains. *_8d_equivalent = SURVEY 8(d)'s bytes for the step-by-step algorithm (all len passes over "
                                 "the decoder weights, which this path does not make) x queries/s: an equivalent-work ratio, not "
                                 "a bandwidth. traffic = (2 FETCH_SIZE + WRITE_SIZE) x 1024 summed over every kernel of one search "
                                 "(rocprofv3 --pmc, tools/small_batch_pmc.sh), per search"}}
    return out


def secondary_beam1000(E, synth, ctx, model, trie, dims, dev, L):
    """The reference retrieval script's operating point (full_evaluate_t5seq_aq_encoder.sh:191-199: --topk=1000 --batch_size=1):
    beam 1000 at one query per search (the script's literal flags) and at the batch the evaluate CLI forms by itself
    (ripor_amd.evaluate.search_batch_size: the KV cache of the batch in ~60 % of the free HBM). Per batch size: queries/s, ms per
    search, and an MFMA `roofline` object over the projection GEMMs of one event-timed eager search."""
    from ripor_amd import evaluate as ev
    cfg = type("Cfg", (), dict(num_decoder_layers=dims.num_decoder_layers, num_heads=dims.num_heads, d_kv=dims.d_kv, d_model=dims.d_model,
                               d_ff=dims.d_ff, decoder_vocab_sizes=list(dims.decoder_vocab_sizes)))
    auto = ev.search_batch_size(cfg, 1, 1000, L, -1, dev)
    out = {"workload": f"t5-base dims, {trie.N}-doc trie, beams=1000, len={L}", "unit": "queries/s", "cli_automatic_batch": auto}
    for tag, Q, steps in (("batch1", 1, 6), ("cli_batch", auto, 2)):
        batches = _query_batches(synth, dims, Q, 2, dev, seed=606 + Q)
        (dt, r), fb = _leftover_guard(ctx, lambda: _time_search(E, model, trie, batches, 1000, L, steps, warmup=1))
        forks = ctx.last_fork_stats()
        lanes = 0 < ctx.lane_split() <= Q * 1000 and Q >= 2
        st = _profiled(ctx, lambda: E.search(model, trie, batches[0][0], batches[0][1], 1000, L, use_graph=False))
        roof = _gemm_roofline(st, PEAK_F16_MFMA_TFLOPS / 3.0, 0.5 if lanes else 1.0, "rpr::gemm_h2_pp_kernel (+ small-tile kernels)",
                              "algorithmic 2MNK flops of every projection launch of one search / their summed event durations; 3 f16 "
                              "MFMAs per product -> peak 2500/3 TF/s, halved per launch when the search runs as two CU-masked lanes")
        out[tag] = {"queries_per_search": Q, "value": Q / dt, "ms_per_search": dt * 1e3, "ms_per_query": dt * 1e3 / Q, "steps": steps,
                    "forks_last_search": forks, "leftover_fallback_taken": fb, "lanes": 2 if lanes else 1, "roofline": roof,
                    "select_ms_event_timed": st["select"]["total_ms"], "select_launches": st["select"]["launches"],
                    "valid_leaves": f"{int((r.row_hi > r.row_lo).sum().item())}/{Q * 1000}"}
    return out


def secondary_rankdata_ref_flags(E, synth, ctx, model, trie, dims, dev, steps=8):
    """SURVEY §8 row f2 at the reference script's literal flags (full_evaluate_t5seq_aq_encoder.sh:117-147: --topk=100
    --batch_size=4, max_new_token 4 / 8 / 16): four queries per search, beam 100. `f2` above is the same caller at the batch the
    CLI forms by itself."""
    out = {"workload": f"t5-base dims, {trie.N}-doc trie, beams=100, 4 queries per search, prefix search", "unit": "queries/s"}
    batches = _query_batches(synth, dims, 4, 4, dev, seed=707)
    for Lp in (4, 8, 16):
        (dt, r), fb = _leftover_guard(ctx, lambda: _time_search(E, model, trie, batches, 100, Lp, steps, warmup=2))
        st = _profiled(ctx, lambda: E.search(model, trie, batches[0][0], batches[0][1], 100, Lp, use_graph=False))
        roof = _gemm_roofline(st, PEAK_F16_MFMA_TFLOPS / 3.0, 1.0, "rpr::gemm_h2_wsplit_kernel / gemm_h2_dma_kernel (400 rows per step)",
                              "algorithmic 2MNK flops of every projection launch of one search / their summed event durations; 3 f16 "
                              "MFMAs per product -> peak 2500/3 TF/s; at 400 rows these launches are bound by the weight stream and "
                              "the launch chain, not by the matrix pipe")
        out[f"len{Lp}"] = {"value": 4 / dt, "ms_per_search": dt * 1e3, "forks_last_search": ctx.last_fork_stats(),
                           "leftover_fallback_taken": fb, "launches_per_search": sum(int(v["launches"]) for v in st.values()),
                           "select_ms_event_timed": st["select"]["total_ms"], "roofline": roof}
    return out


def secondary_heavy_tail(E, synth, ctx, trie, dims, dev, B, L, Q, steps=4, target=1e5):
    """the headline configuration on weights with the activation statistics of trained T5 checkpoints
    (synth.make_state_dict(outliers=1e5): residual channels of ~5e4 from the embedding to the last block) instead of N(0, sigma)
    weights: queries/s, whether any activation left the f16 planes, and how many of the timed batches the guard of the
    reference-shaped entry point would have repeated in exact fp32."""
    sd = synth.make_state_dict(dims, outliers=target, logit_scale=3.0)
    model = E.DeviceModel(ctx, sd, dims)
    batches = _query_batches(synth, dims, Q, 2, dev, seed=505)
    repeated = 0
    ctx.status(clear=True)
    for b in batches:                                     # warm-up (graphs) + one flag check per batch
        E.search(model, trie, b[0], b[1], B, L)
        if ctx.status(clear=True) & 1:
            repeated += 1
    (dt, r), fb = _leftover_guard(ctx, lambda: _time_search(E, model, trie, batches, B, L, steps, warmup=0))
    sat = bool(ctx.status(clear=True) & 1)
    out = {"value": Q / dt, "unit": "queries/s", "ms_per_step": dt * 1e3, "steps": steps,
           "workload": f"t5-base dims with heavy-tailed weights (outlier channels ~{target:g}), {trie.N}-doc trie, beams={B}, len={L}, {Q} queries/step",
           "saturated": sat, "model_f32_only": bool(model.f32_only), "batches_checked": len(batches),
           "batches_the_guard_would_repeat_in_fp32": repeated, "leftover_fallback_taken": fb,
           "valid_leaves": f"{int((r.row_hi > r.row_lo).sum().item())}/{Q * B}"}
    del model
    torch.cuda.empty_cache()
    return out


def secondary_f2(E, synth, ctx, model, trie, dims, dev, queries=214, steps=5):
    """the training-data generation callers (evaluate.py:134-178; full_evaluate_t5seq_aq_encoder.sh:117-147):
    the same search at max_new_token 4 / 8 / 16 with topk = 100."""
    out = {"workload": f"t5-base dims, {trie.N}-doc trie, beams=100, {queries} queries/step, prefix search", "unit": "queries/s"}
    batches = _query_batches(synth, dims, queries, 2, dev, seed=202)
    for Lp in (4, 8, 16):
        (dt, r), fb = _leftover_guard(ctx, lambda: _time_search(E, model, trie, batches, 100, Lp, steps))
        out[f"len{Lp}"] = {"value": queries / dt, "ms_per_step": dt * 1e3, "forks_last_step": ctx.last_fork_stats(),
                           "leftover_fallback_taken": fb}
    return out


def secondary_skew(E, synth, ctx, model, dims, dev, docs, B, L, V, queries, steps=5):
    """The skewed trie of SURVEY §8(d): Zipf s = 1.0 on levels 1-3 — residual-quantiser codes are imbalanced
    (aq_preprocess/create_customized_smtid_file.py:33-59) — plus 10 % of the docs sharing their smtid with another doc
    (evaluate.py:439-446 keeps every docid of an smtid). Popular prefixes stay dense for longer: the automatic forks move
    from {4, 6} to {5, 7} and more queries walk on after the first one."""
    t0 = time.time()
    codes = synth.make_codes_fast(docs, L, V, seed=synth.SEED + 1, zipf=1.0)
    ndup = docs // 10
    src = (synth.hash_u64(f"bench_dup/{docs}", ndup, synth.SEED) % np.uint64(docs)).astype(np.int64)
    codes[docs - ndup:] = codes[src]
    trie = E.DeviceTrie.from_codes(ctx, codes, V)
    frac = E.trie_single_frac(codes[:: max(1, docs // 2_000_000)], L) if docs > 4_000_000 else E.trie_single_frac(codes, L)
    del codes
    t_trie = time.time() - t0
    batches = _query_batches(synth, dims, queries, 2, dev, seed=303)
    (dt, r), fb = _leftover_guard(ctx, lambda: _time_search(E, model, trie, batches, B, L, steps))
    forks = ctx.last_fork_stats()
    ctx.profile_reset(); ctx.profile_enable(True)
    E.search(model, trie, batches[0][0], batches[0][1], B, L)
    torch.cuda.synchronize()
    st = ctx.profile_get(); ctx.profile_enable(False)
    tot = sum(v["total_ms"] for v in st.values())
    ok = int((r.row_hi > r.row_lo).sum().item())
    multi = int((r.row_hi - r.row_lo > 1).sum().item())
    out = {"workload": f"t5-base dims, {docs}-doc SKEWED trie (Zipf s=1.0 codes on levels 1-3 as SURVEY 8d asks, 10 % duplicated smtids), "
                       f"beams={B}, len={L}, {queries} queries/step",
           "value": queries / dt, "unit": "queries/s", "ms_per_step": dt * 1e3, "steps": steps,
           "forks_last_step": forks, "leftover_fallback_taken": fb,
           "select_share_of_kernel_time": st["select"]["total_ms"] / max(1e-9, tot),
           "kernel_breakdown_ms": {k: round(v["total_ms"], 3) for k, v in st.items()},
           "valid_leaves": f"{ok}/{queries * B}", "returned_smtids_with_several_docs": multi, "trie_build_s": round(t_trie, 1),
           "single_sequence_node_share_by_depth": [round(float(x), 4) for x in frac[:10]]}
    del trie
    return out


def secondary_v1024(E, synth, ctx, dev, docs, B, queries, steps=5):
    """RIPOR's 16 x 1024 codebook variant (reference full_16_1024_scripts/full_evaluate_t5seq_aq_encoder.sh:19-22: M = 16,
    nbits = 10): t5-base dims, smtids of 16 tokens over 1024-entry codebooks, the same number of docs and queries."""
    L, V = 16, 1024
    dims = synth.t5_base_dims(L=L, V=V)
    t0 = time.time()
    model = E.DeviceModel(ctx, synth.make_state_dict(dims), dims)
    codes = synth.make_codes_fast(docs, L, V, seed=synth.SEED + 2)
    trie = E.DeviceTrie.from_codes(ctx, codes, V)
    del codes
    t_setup = time.time() - t0
    batches = _query_batches(synth, dims, queries, 2, dev, seed=505)
    (dt, r), fb = _leftover_guard(ctx, lambda: _time_search(E, model, trie, batches, B, L, steps))
    forks = ctx.last_fork_stats()
    ok = int((r.row_hi > r.row_lo).sum().item())
    lanes = 0 < ctx.lane_split() <= queries * B
    st = _profiled(ctx, lambda: E.search(model, trie, batches[0][0], batches[0][1], B, L))
    out = {"workload": f"t5-base dims, {docs}-doc trie of 16 x 1024 codes, beams={B}, len={L}, {queries} queries/step",
           "value": queries / dt, "unit": "queries/s", "ms_per_step": dt * 1e3, "steps": steps, "forks_last_step": forks,
           "leftover_fallback_taken": fb, "valid_leaves": f"{ok}/{queries * B}", "setup_s": round(t_setup, 1),
           "roofline": _gemm_roofline(st, PEAK_F16_MFMA_TFLOPS / 3.0, 0.5 if lanes else 1.0, "rpr::gemm_h2_pp_kernel (+ small-tile kernels)",
                                      "2MNK flops of every projection launch of one step / summed event durations; 3 f16 MFMAs per product"),
           "kernel_breakdown_ms": {k: round(v["total_ms"], 3) for k, v in st.items()}}
    del model, trie
    return out


def secondary_train_step(E, synth, ctx, dev, world, rank, bz=128, L=32, steps=5, precision=None):
    """BASELINE config 5 / SURVEY §8 row f4: one optimisation step of the prefix-oriented ranking fine-tune
    (forward + backward + gradient all-reduce over the ranks + clip + AdamW), t5-base dims, bz examples per GPU."""
    import torch.distributed as dist
    V = 256
    dims = synth.t5_base_dims(L=L, V=V)
    model = E.DeviceModel(ctx, synth.make_state_dict(dims), dims)
    state = E.TrainState(model)
    ids, mask = synth.make_queries(bz * world, vocab_size=dims.vocab_size, seed=5, mean_len=16, std_len=5, min_len=6, max_len=64)
    ids, mask = ids[rank::world], mask[rank::world]
    Lq = (ids.shape[1] + 7) // 8 * 8
    ids = np.pad(ids, ((0, 0), (0, Lq - ids.shape[1]))); mask = np.pad(mask, ((0, 0), (0, Lq - mask.shape[1])))
    codes = synth.make_codes(2 * bz, L, V, seed=5).astype(np.int64).reshape(2, bz, L).transpose(1, 0, 2).copy()
    prefix = [L, 4, 8, 16][: {8: 2, 16: 3, 32: 4}[L]]
    tp = torch.from_numpy(np.stack([synth.uniform_f32(f"tb/p{k}", (bz,), 30.0) for k in prefix]))
    tn = torch.from_numpy(np.stack([synth.uniform_f32(f"tb/n{k}", (bz,), 30.0) for k in prefix]))
    ids_t, mask_t, codes_t = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev), torch.from_numpy(codes).to(dev)
    saved = ctx.get_precision()
    if precision:
        ctx.set_precision(precision)
    try:
        def step():
            return E.train_step(model, state, ids_t, mask_t, codes_t, tp, tn, prefix, lr=1e-6)

        first = step()
        step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            last = step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = (time.perf_counter() - t0) / steps
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        prec = ctx.get_precision()
        st = _profiled(ctx, step)
        peak = {"bf16": PEAK_F16_MFMA_TFLOPS, "f16x2": PEAK_F16_MFMA_TFLOPS / 3.0, "f32": PEAK_F32_MFMA_TFLOPS}[prec]
        roof = _gemm_roofline(st, peak, 1.0, "rpr::gemm_h2_pp_kernel / gemm_h2_dma_kernel" + (" <BF16>" if prec == "bf16" else ""),
                              "2MNK flops of every GEMM launch of one optimisation step (forward with saved activations, input and "
                              "weight gradients) / their summed event durations (side-stream launches overlap the main stream's, so "
                              "the sum can exceed the step's wall clock); peak = dense MFMA peak of the arithmetic "
                              "(bf16: one MFMA per product; f16x2: three)")
    finally:
        ctx.set_precision(saved)
    out = {"workload": f"lng_knp margin-MSE fine-tune step (forward + backward + gradient all-reduce + clip + AdamW), t5-base dims, "
                       f"bz={bz}/GPU, smtid len {L}, queries padded to {int(Lq)}",
           "value": world * bz / dt, "unit": "examples/s", "ms_per_step": dt * 1e3, "steps": steps, "n_gpus": world,
           "gemm_arithmetic": prec, "roofline": roof, "loss_first": [float(x) for x in first], "loss_last": [float(x) for x in last],
           "allreduce_bytes_per_step_per_rank": int(state.total * 4) if world > 1 else 0,
           "allreduce": E.allreduce_mode() if world > 1 else "none (1 rank)", "params": int(state.total)}
    del model, state
    return out


def launch_command(gpus, env, argv):
    """The N>1 contract is one process per GPU under torch.distributed.run. When --gpus N > 1 is given without a
    launcher (no WORLD_SIZE in the environment), return the command that re-runs this script as N ranks on
    127.0.0.1 with a free port; None when already launched or N == 1."""
    if gpus <= 1 or "WORLD_SIZE" in env:
        return None
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(argv[0])] + list(argv[1:])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=None, help="queries in flight per step per GPU; default 2150 = two lanes of 1075 queries (10 750 rows = 42 row tiles of 256: 126 / 378 / 504 GEMM tiles = whole rounds of a lane's 128 CUs), 2176 with --no-lanes (85 row tiles: 255 / 765 / 1020 tiles on 256 CUs); 4300 / 6450 (two / three times the rows: 82 / 123 GiB of workspace) measured +1.0 / +1.7 % same-box in round 5 and are not the default: with the exact-fp32 leg, the plain-loop parity test of this configuration and the secondary legs beside it 6450 exhausts the 288 GB")
    ap.add_argument("--beams", type=int, default=10)
    ap.add_argument("--len", type=int, default=32, dest="L")
    ap.add_argument("--docs", type=int, default=MSMARCO_DOCS)
    ap.add_argument("--model", default="t5-base", choices=["t5-base", "t5-large"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-exact-fp32", action="store_true", help="skip the secondary exact-fp32 timing")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--train-steps", type=int, default=5, dest="train_steps", help="timed steps of the secondary train legs")
    ap.add_argument("--train-bz", type=int, default=128, dest="train_bz", help="examples per GPU and step of the secondary train legs")
    ap.add_argument("--secondary", default="train,config4,f2,rankdata_ref_flags,skew,latency,beam1000,small_batch,heavy_tail,v1024",
                    help="comma list of secondary legs to append to the JSON line (train = BASELINE config 5 step, config4 = "
                         "t5-large beam 100, f2 = prefix search at topk 100, skew = clustered trie, latency = one query at beams 10 and "
                         "1000, t5_3b = opt-in: t5-3b dims at beam 10); '' = none. config4 / f2 / skew / latency run at --gpus 1 only, train on every rank (its gradient all-reduce is the RCCL leg)")
    ap.add_argument("--log-softmax", action="store_true", dest="log_softmax",
                    help="apply_log_softmax_for_scores=True (reference generation.py:453-455; not the headline configuration)")
    ap.add_argument("--no-lanes", action="store_true",
                    help="one stream for the whole batch instead of two half batches on two CU-masked streams")
    ap.add_argument("--forced-tail", type=int, default=2, choices=[0, 1, 2], dest="forced_tail",
                    help="0 = plain step-by-step loop, 1 = forced tail, exact on the device, 2 = optimistic (default; "
                         "falls back to 1 if a query was left unforced)")
    ap.add_argument("--precision", default="f16x2", choices=["f16x2", "f32"],
                    help="GEMM arithmetic: f16x2 = fp32 operands as two f16 planes, 3 f16 MFMAs per product "
                         "(fp32-equivalent to ~2^-22); f32 = exact fp32 MFMA")
    args = ap.parse_args()

    relaunch = launch_command(args.gpus, os.environ, sys.argv)
    if relaunch is not None:   # `python bench.py --gpus N` without a launcher: become N ranks (one per GPU)
        log(f"[bench] --gpus {args.gpus} without WORLD_SIZE: re-launching as " + " ".join(relaunch))
        os.execvp(relaunch[0], relaunch)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:     # never print an N=1 number under --gpus 8 (or the other way round)
        raise SystemExit(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: refusing to run a mislabelled job")
    import torch.distributed as dist
    backend = os.environ.get("RPR_BENCH_BACKEND", "nccl")   # "nccl" = RCCL on ROCm; tests use gloo on CPU for the launcher
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a job that has a GPU per rank runs over RCCL, whatever the environment says: the scaling bench must never fall to gloo
        # silently (RPR_BENCH_BACKEND=gloo is the hook of the one-GPU rehearsals in tests/)
        if backend != "nccl" and torch.cuda.device_count() >= world:
            raise SystemExit(f"[bench] {torch.cuda.device_count()} GPUs visible for {world} ranks but RPR_BENCH_BACKEND={backend}: "
                             "refusing to measure the multi-GPU path over anything but RCCL ('nccl')")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    if world > 1 and torch.cuda.device_count() >= world:
        assert dist.get_backend() == "nccl", dist.get_backend()
    if os.environ.get("RPR_BENCH_LAUNCH_ONLY"):   # launcher self-test (tests/test_dist_gloo.py): rendezvous, report, exit
        if world > 1:
            t = torch.tensor([rank], dtype=torch.int64)
            lst = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(lst, t)
            if rank == 0:
                print(json.dumps({"n_gpus": args.gpus, "rccl_world_size": dist.get_world_size(),
                                  "ranks_seen": [int(x) for x in lst]}), flush=True)
            dist.barrier()
            dist.destroy_process_group()
        else:
            print(json.dumps({"n_gpus": 1, "rccl_world_size": 1, "ranks_seen": [0]}), flush=True)
        return
    # RPR_BENCH_DEVICE: test hook (tests/test_gpu_cli.py runs two gloo ranks on the one GPU of the test box to exercise the
    # N > 1 code path end to end: sharding, the result gather, the bucketed gradient exchange of the train leg)
    dev = torch.device("cuda", int(os.environ.get("RPR_BENCH_DEVICE", local_rank)))
    torch.cuda.set_device(dev)

    from ripor_amd import engine as E
    from ripor_amd.dataset.sharding import shard_indices
    from ripor_amd.utils import synth

    Q, B, L, K, W = args.batch, args.beams, args.L, args.steps, args.warmup
    dims = synth.t5_base_dims(L=L) if args.model == "t5-base" else synth.t5_large_dims(L=L)
    V = dims.decoder_vocab_sizes[0]
    t0 = time.time()
    sd = synth.make_state_dict(dims)
    ctx = E.Context.get(dev.index)
    ctx.set_precision(args.precision)
    # forced-tail evaluation: 2 = optimistic (no stage after the last fork; a query left unforced there raises the
    # sticky TAIL_LEFTOVER word, checked after the timed region — the number is then void and the run is repeated in
    # the exact mode 1), 1 = exact on the device, 0 = plain step-by-step loop
    ctx.set_forced_tail(args.forced_tail)
    if args.no_lanes:
        ctx.set_lane_split(0)
    if args.batch is None:
        args.batch = 2150 if ctx.lane_split() > 0 else 2176
    Q = args.batch
    lane_min_rows = ctx.lane_split()
    lanes_on = 0 < lane_min_rows <= Q * args.beams and Q >= 2
    model = E.DeviceModel(ctx, sd, dims)
    log(f"[bench r{rank}] weights ({sum(v.size for v in sd.values()) / 1e6:.1f} M params) on device in {time.time() - t0:.1f}s")
    t0 = time.time()
    codes = synth.make_codes_fast(args.docs, L, V)
    trie = E.DeviceTrie.from_codes(ctx, codes, V)
    del codes
    log(f"[bench r{rank}] trie of {trie.N:,} docs built in {time.time() - t0:.1f}s")

    # queries: a pool of MSMARCO-dev size, sharded like DistributedSampler(shuffle=False); every step
    # takes the next Q queries of this rank's shard (wrapping), padded to the batch maximum.
    pool_ids, pool_mask = synth.make_queries(6980, vocab_size=dims.vocab_size)
    shard = shard_indices(6980, world, rank)
    # One padded length for every step of this rank (the longest query of its shard, bucketed to 8): the
    # hipGraph captured in the warmup is then the one replayed in the timed steps. Padding is free on the
    # device (the encoder runs on packed rows), it only sizes the mask and the cross-attention LDS.
    sels = [[shard[(step * Q + i) % len(shard)] for i in range(Q)] for step in range(W + K + 1)]
    lq = max(int(pool_mask[sel].sum(1).max()) for sel in sels)
    lq = (lq + 7) // 8 * 8
    batches = []
    for sel in sels:
        ids, mask = pool_ids[sel], pool_mask[sel]
        ids = np.pad(ids, ((0, 0), (0, max(0, lq - ids.shape[1]))))[:, :lq]
        mask = np.pad(mask, ((0, 0), (0, max(0, lq - mask.shape[1]))))[:, :lq]
        batches.append((torch.from_numpy(ids).to(dev, torch.int32), torch.from_numpy(mask).to(dev, torch.int32), lq))
    mean_len = float(pool_mask.sum(1).mean())

    def run_step(i):
        ids, mask, _ = batches[i]
        return E.search(model, trie, ids, mask, B, L, use_graph=not args.no_graph, apply_log_softmax_for_scores=args.log_softmax)

    gathered = {"bytes": 0}

    def timed_region():
        for i in range(W):
            res = run_step(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t_start = time.perf_counter()
        results = []
        for i in range(K):
            results.append(run_step(W + i))
        if world > 1:  # the path's only collective: gather the ranked results (run.json merge)
            tok = torch.stack([r.tokens for r in results])
            sc = torch.stack([r.scores for r in results])
            tok_all = torch.empty((world * tok.shape[0],) + tuple(tok.shape[1:]), dtype=tok.dtype, device=dev)
            sc_all = torch.empty((world * sc.shape[0],) + tuple(sc.shape[1:]), dtype=sc.dtype, device=dev)
            dist.all_gather_into_tensor(tok_all, tok)
            dist.all_gather_into_tensor(sc_all, sc)
            gathered["bytes"] = int(tok_all.numel() * tok_all.element_size() + sc_all.numel() * sc_all.element_size())
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t_start
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())

        # Second timed loop, same K steps, with the boundary's PCIe legs inside (SURVEY §8d wording of the metric): ids and
        # mask start in pinned host memory, results end in pinned host memory. Reported as value_pcie_inclusive; `value`
        # stays the resident-input rate.
        host_in = [(b[0].cpu().pin_memory(), b[1].cpu().pin_memory()) for b in batches[W:W + K]]
        host_out = None
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t_pcie = time.perf_counter()
        for i in range(K):
            ids_d = host_in[i][0].to(dev, non_blocking=True)
            mask_d = host_in[i][1].to(dev, non_blocking=True)
            r = E.search(model, trie, ids_d, mask_d, B, L, use_graph=not args.no_graph, apply_log_softmax_for_scores=args.log_softmax)
            if host_out is None:
                host_out = [torch.empty(x.shape, dtype=x.dtype).pin_memory() for x in (r.tokens, r.scores, r.row_lo, r.row_hi)]
            for dst, src in zip(host_out, (r.tokens, r.scores, r.row_lo, r.row_hi)):
                dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed_pcie = time.perf_counter() - t_pcie
        if world > 1:
            t = torch.tensor([elapsed_pcie], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed_pcie = float(t.item())
        status_flags = ctx.status(clear=True)   # sticky saturation / empty-query word over everything run so far
        return elapsed, elapsed_pcie, status_flags, results, (tok if world > 1 else None), (sc if world > 1 else None)

    distinct_shards = 1
    if world > 1:   # every rank works on its own shard of the query pool: the first timed query of each must differ
        first = torch.tensor([sels[W][0]], dtype=torch.int64, device=dev)
        lst = [torch.zeros_like(first) for _ in range(world)]
        dist.all_gather(lst, first)
        distinct_shards = len({int(x.item()) for x in lst})
    elapsed, elapsed_pcie, status_flags, results, tok, sc = timed_region()
    tail_mode = args.forced_tail
    leftover = bool(status_flags & 4)
    if world > 1:   # every rank must take the same path
        t = torch.tensor([1 if leftover else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        leftover = bool(t.item())
    if leftover and tail_mode == 2:
        # optimistic forced tail: a query was still unforced at the last fork, its outputs are unspecified -> the timed
        # region is void; repeat it in the exact mode
        log("[bench] TAIL_LEFTOVER raised in optimistic mode: repeating the timed region in the exact forced-tail mode")
        tail_mode = 1
        ctx.set_forced_tail(1)
        elapsed, elapsed_pcie, status_flags, results, tok, sc = timed_region()

    fork_stats = ctx.last_fork_stats()   # of the last search (the PCIe-inclusive loop's last step: same shapes)
    # sanity on the timed outputs: every returned smtid of the last step is a trie leaf range
    last = results[-1]
    n_leaf = int((last.row_hi > last.row_lo).sum().item())
    if rank == 0:
        log(f"[bench] last step: {n_leaf}/{Q * B} returned smtids are valid trie leaves; "
            f"workspace {ctx.workspace_bytes() / 2**30:.2f} GiB")

    out = None
    if rank == 0:
        lq_used = batches[W][2]
        ms_per_step = elapsed / K * 1e3
        value = world * Q * K / elapsed
        # algorithmic work is counted on the queries' own tokens (mean of the timed batches), not on the padding
        lq_alg = float(np.mean([float(batches[W + i][1].sum().item()) / Q for i in range(K)]))
        abytes = algorithmic_bytes_per_query(dims, Q, B, L, lq_alg)
        aflops = algorithmic_flops_per_query(dims, B, L, lq_alg)
        out = {
            "metric": "queries/sec, t5-base beam=10 len=32 over 8.8M-doc trie (constrained beam search)",
            "value": value, "unit": "queries/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "f32" else "f32 via f16x2-split MFMA (fp32 accumulate)", "data": "synthetic",
            "value_pcie_inclusive": world * Q * K / elapsed_pcie,
            "rccl_world_size": dist.get_world_size() if world > 1 else 1,
            "dist_backend": dist.get_backend() if world > 1 else None,
            "gather_bytes_per_rank": int(tok.numel() * tok.element_size() + sc.numel() * sc.element_size()) if world > 1 else 0,
            "gathered_bytes_total": gathered["bytes"], "distinct_shards": distinct_shards,
            "saturated": bool(status_flags & 1), "model_f32_only": bool(model.f32_only),
            "forced_tail": {"mode": {0: "off (step-by-step loop)", 1: "exact", 2: "optimistic"}[tail_mode],
                            "forks_last_step": fork_stats,
                            "leftover_fallback_taken": bool(leftover and args.forced_tail == 2),
                            "note": "forks: depth at which queries whose beams can no longer be pruned leave the sequential "
                                    "steps (forced = scored by one teacher-forced tail pass, left = walked on); depths chosen "
                                    "from the trie statistics (rpr_trie_single_frac)"},
            "config": {"workload": f"{args.model} dims, {trie.N}-doc synthetic 32x256 docid trie, beams={B}, len={L}, "
                                   f"{Q} queries/step/GPU (MSMARCO-dev-shaped, mean {mean_len:.1f} tokens, padded to {lq_used})",
                       "queries_per_step_per_gpu": Q, "beams": B, "len": L, "docs": trie.N, "enc_len_padded": lq_used,
                       "parallelism": f"query-sharded x{world}, replicated weights+trie, final RCCL all_gather",
                       "hipgraph": not args.no_graph, "gemm_precision": args.precision,
                       "lanes": ("2 half batches on 2 HIP streams confined to half of the CUs each (rpr_set_lane_split)"
                                 if lanes_on else "1 (whole batch on one stream)")},
            "algorithmic": {"bytes_per_query": abytes, "flops_per_query": aflops, "tokens_per_query": lq_alg,
                            "formula": "SURVEY.md §8(d): the KV-cached step-by-step algorithm",
                            "hbm_frac_whole_step": abytes * Q / (ms_per_step * 1e-3) / (PEAK_HBM_TBS * 1e12),
                            "tflops_whole_step": aflops * Q / (ms_per_step * 1e-3) / 1e12,
                            "frac_of_f16x3_mfma_peak_whole_step": aflops * Q / (ms_per_step * 1e-3) / (PEAK_F16_MFMA_TFLOPS / 3.0 * 1e12),
                            "vs_native_f32_mfma_peak": aflops * Q / (ms_per_step * 1e-3) / (PEAK_F32_MFMA_TFLOPS * 1e12),
                            "note": "vs_native_f32_mfma_peak > 1: the 3-pass f16 split runs fp32-equivalent products faster than "
                                    "v_mfma_f32_32x32x2_f32 could at its 157 TF/s peak"},
        }
        fk = out["forced_tail"].get("forks_last_step") or []
        if args.forced_tail and fk:
            fb, ff = algorithmic_forced_tail_per_query(dims, Q, B, L, lq_alg, fk[0]["depth"])
            out["algorithmic"]["forced_tail"] = {
                "fork_depth": fk[0]["depth"], "bytes_per_query": fb, "flops_per_query": ff,
                "hbm_frac_whole_step": fb * Q / (ms_per_step * 1e-3) / (PEAK_HBM_TBS * 1e12),
                "note": "the algorithm the timed region runs (first fork only): fewer bytes than §8(d) — no per-step self-KV "
                        "re-reads past the fork, the decoder weights streamed fork_depth + 1 times instead of len times"}
        log(f"[bench] timed region done: {value:.1f} queries/s, {ms_per_step:.1f} ms/step")
        if not args.no_roofline:
            def profiled_step():
                ctx.profile_reset()
                ctx.profile_enable(True)
                run_step(W)  # one eager step, hipEvents around every launch on the stream it is launched on
                torch.cuda.synchronize()
                st = ctx.profile_get()
                ctx.profile_enable(False)
                return st

            def gemm_roofline(stats, cu_fraction, with_traffic=True):
                g = stats["gemm"]           # launches of the dominant kernel only (256x256 ping-pong / fp32 MFMA kernel);
                gs = stats["gemm_small"]    # the few small-tile launches (encoder tail, logits of step 0) are listed apart
                small_only = g["launches"] == 0      # a handful of rows in flight: every GEMM runs on the small-tile kernels
                if small_only:
                    g, gs = gs, g
                ach = g["flops"] / max(1e-9, g["total_ms"] * 1e-3) / 1e12
                if args.precision == "f32":
                    kname, full_peak, note = "rpr::gemm_f32_kernel", PEAK_F32_MFMA_TFLOPS, "exact fp32 MFMA (v_mfma_f32_32x32x2_f32)"
                else:
                    kname, full_peak = "rpr::gemm_h2_pp_kernel", PEAK_F16_MFMA_TFLOPS / 3.0
                    if small_only:
                        kname = "rpr::gemm_h2_dma_kernel / rpr::gemm_h2_skinny_kernel"
                    note = ("achieved counts algorithmic 2MNK flops; the kernel issues 3 f16 MFMAs per product "
                            "(hi*hi + hi*lo + lo*hi), so peak = 2500 TF/s dense f16 / 3 at the nominal 2.4 GHz; the in-kernel "
                            "s_memtime/s_memrealtime trace (tools/attic/gemm_trace_pp.py) shows the chip sustaining 1.64-1.76 GHz "
                            "under this kernel (DVFS), i.e. ~0.6 of the peak at the sustained clock")
                peak = full_peak * cu_fraction
                if cu_fraction != 1.0:
                    note += (f"; LANES: every launch of the timed region is one half batch on a HIP stream confined to "
                             f"{cu_fraction:.2f} of the CUs (hipExtStreamCreateWithCUMask), the other half batch runs beside it "
                             f"on the other CUs, so the peak of a launch is {cu_fraction:.2f} x the chip's "
                             f"({full_peak:.1f} TF/s); roofline_unsplit = the same kernel launched on the whole chip "
                             f"(bench.py --no-lanes)")
                traffic, traffic_src = None, None
                pmc_path = os.path.join(REPO, "profiles", "latest_hbm_pmc.json")
                if with_traffic and os.path.exists(pmc_path):  # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh
                    try:
                        pmc = json.load(open(pmc_path))
                        key = [k for k in pmc["FETCH_SIZE"] if ("gemm_h2_pp_kernel" if args.precision != "f32" else "gemm_f32_kernel<128, 128") in k]
                        key.sort(key=lambda k: -pmc["FETCH_SIZE"][k]["launches"])   # the instantiation that ran most
                        if key:
                            # KB per launch; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 (MI355X_MICROARCH.md §HBM)
                            traffic = (2.0 * pmc["FETCH_SIZE"][key[0]]["mean"] + pmc["WRITE_SIZE"][key[0]]["mean"]) * 1024.0
                            pmc_build = pmc.get("_meta", {}).get("build")
                            traffic_src = ("profiles/latest_hbm_pmc.json: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate "
                                           "passes, mean per launch of " + key[0] + ", bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024; "
                                           f"counters of build {pmc_build if pmc_build else 'r04n (no build id recorded)'}, "
                                           f"this run is build {BUILD_ID}")
                            # the counters come from an earlier profiled run of tools/profile_round.sh: refuse them when that
                            # run's launch count per step no longer matches this build's (kernels changed since)
                            kn = "gemm_h2_pp_kernel" if args.precision != "f32" else "gemm_f32_kernel"
                            n_pmc = sum(v["launches"] for k, v in pmc["FETCH_SIZE"].items() if isinstance(v, dict) and kn in k)
                            n_pmc /= float(pmc.get("_meta", {}).get("steps_in_pmc_pass", 1))
                            if abs(n_pmc - g["launches"]) > 0.5:
                                traffic_src = (f"stale: profiles/latest_hbm_pmc.json has {n_pmc:g} launches of the kernel per step, "
                                               f"this run {g['launches']}; re-run tools/profile_round.sh")
                                traffic = None
                    except Exception:
                        pass
                src = None
                csv_path = os.path.join(REPO, "profiles", "latest_kernel_stats.csv")
                if cu_fraction != 1.0 or not lanes_on:
                    try:   # rocprofv3 --kernel-trace --stats of `bench.py --no-roofline --secondary ""` (tools/profile_round.sh)
                        import csv as _csv
                        want = kname.split("::")[-1].split(" ")[0]
                        rows = [r for r in _csv.DictReader(open(csv_path)) if want in r["Name"]]
                        rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
                        if rows:
                            us = float(rows[0]["AverageNs"]) / 1e3
                            mine = g["total_ms"] * 1e3 / max(1, g["launches"])
                            try:
                                trace_build = json.load(open(os.path.join(REPO, "profiles", "latest_kernel_stats.meta.json")))["build"]
                            except Exception:
                                trace_build = "r04r (no build id recorded)"
                            src = {"file": "profiles/latest_kernel_stats.csv", "row": rows[0]["Name"].split("(")[0],
                                   "trace_build": trace_build, "this_build": BUILD_ID,
                                   "calls": int(rows[0]["Calls"]), "rocprof_avg_launch_us": us,
                                   "this_run_over_rocprof": mine / us,
                                   "note": "rocprofv3 --kernel-trace --stats of the graph-replayed timed region of an earlier run of "
                                           "the same build (bench.py --no-roofline --secondary ''); the eager event-timed average "
                                           "above should agree within a few percent (boxes differ by +-2 %)"}
                    except Exception:
                        pass
                return {"kernel": kname, "bound": "mfma", "achieved": ach,
                        "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "cu_fraction": cu_fraction,
                        "source_profile": src,
                        "peak_whole_chip": full_peak, "note": note,
                        "traffic": traffic, "traffic_source": traffic_src,
                        "algorithmic_bytes_per_launch": g["bytes"] / max(1, g["launches"]),
                        "avg_launch_us": g["total_ms"] * 1e3 / max(1, g["launches"]),
                        "launches_per_step": g["launches"],
                        "flops_per_launch": g["flops"] / max(1, g["launches"]),
                        "other_gemm_launches": {"launches_per_step": gs["launches"], "total_ms": gs["total_ms"],
                                                "tflops": gs["flops"] / max(1e-9, gs["total_ms"] * 1e-3) / 1e12}}

            stats = profiled_step()     # the timed configuration (two lanes when the batch is split)
            log("[bench] roofline leg done")
            out["roofline"] = gemm_roofline(stats, 0.5 if lanes_on else 1.0)
            out["lanes"] = 2 if lanes_on else 1
            # consistency of the event-timed pass with the timed region: a lane executes its launches back to back, so the
            # event durations of everything it ran must add up to about one step of wall clock
            busy = sum(v["total_ms"] for v in stats.values()) / (2 if lanes_on else 1)
            out["roofline"]["lane_time_check"] = {
                "sum_of_event_durations_per_lane_ms": busy, "ms_per_step_timed_region": ms_per_step,
                "ratio": busy / ms_per_step,
                "dominant_kernel_ms_per_lane": stats["gemm"]["total_ms"] / (2 if lanes_on else 1),
                "note": "eager profile pass (hipEvents around every launch, no synchronisation while enqueuing) vs the "
                        "graph-replayed timed region; ratio ~1 means the launches were timed under the same contention as in "
                        "the timed region"}
            if lanes_on:
                # the same step on one stream with every launch on the whole chip: the kernel's own quality, comparable with
                # earlier rounds; the per-kernel breakdown and the HBM roofline below come from this step (in lane mode two
                # kernels run at any time and their durations add up to about twice the wall clock)
                out["kernel_breakdown_lanes_ms"] = {k: round(v["total_ms"], 3) for k, v in stats.items()}
                ctx.set_lane_split(0)
                stats = profiled_step()
                ctx.set_lane_split(lane_min_rows)
                out["roofline_unsplit"] = gemm_roofline(stats, 1.0, with_traffic=False)
            tot = sum(v["total_ms"] for v in stats.values())
            out["kernel_breakdown_ms"] = {k: round(v["total_ms"], 3) for k, v in stats.items()}
            out["kernel_breakdown_ms"]["sum"] = round(tot, 3)
            sa = stats["dec_self_attn"]
            if sa["total_ms"] > 0:
                # second roofline object: the HBM-bound kernel of the step (same accounting as `roofline`:
                # algorithmic bytes / hipEvent time; traffic from the PMC passes summed over the depth instantiations)
                ach = sa["bytes"] / (sa["total_ms"] * 1e-3) / 1e9
                sa_traffic = None
                try:
                    pmc = json.load(open(os.path.join(REPO, "profiles", "latest_hbm_pmc.json")))
                    ks = [k for k in pmc["FETCH_SIZE"] if "dec_self_attn_fast_kernel" in k or "dec_self_attn_kv3_kernel" in k]
                    tot_b = sum((2.0 * pmc["FETCH_SIZE"][k]["sum"] + pmc["WRITE_SIZE"][k]["sum"]) * 1024.0 for k in ks)
                    n = sum(pmc["FETCH_SIZE"][k]["launches"] for k in ks)
                    sa_traffic = tot_b / n if n else None
                except Exception:
                    pass
                if lanes_on:   # the committed counters are per half-batch launch (lane mode); this object is the unsplit step
                    sa_traffic = None
                out["roofline_hbm"] = {"kernel": "rpr::dec_self_attn_fast_kernel<2|4|6|8>", "bound": "hbm", "achieved": ach,
                                       "peak": PEAK_HBM_TBS * 1e3, "unit": "GB/s", "frac": ach / (PEAK_HBM_TBS * 1e3),
                                       "traffic": sa_traffic,
                                       "algorithmic_bytes_per_launch": sa["bytes"] / max(1, sa["launches"]),
                                       "avg_launch_us": sa["total_ms"] * 1e3 / max(1, sa["launches"]),
                                       "launches_per_step": sa["launches"],
                                       "note": "K/V rows of every beam's ancestry, read once per step; ~6.3 TB/s is what "
                                               "streaming reads reach on this part (MI355X_MICROARCH.md)"
                                               + ("; measured on the unsplit step (whole-chip launches); PMC traffic of "
                                                  "whole-chip launches: profiles/r02f_q2176_hbm_pmc.json" if lanes_on else "")}
                out["self_attn_hbm"] = {"achieved_GBs": ach, "frac_of_8TBs": ach / (PEAK_HBM_TBS * 1e3)}
            # the attention kernels of the step against HBM (same accounting: the library's algorithmic bytes per class —
            # q / k / v rows read once, planes written once — over the hipEvent time of the unsplit step's launches)
            hk = {}
            for cls, label in (("tail_self_attn", "rpr::tail_self_attn_mfma_v2_kernel"),
                               ("dec_cross_attn", "rpr::tail_cross_attn_mfma_v2_kernel + step_cross_attn_mfma16_kernel"),
                               ("enc_attn", "rpr::enc_attn_mfma_v2_kernel")):
                v = stats.get(cls)
                if v and v["total_ms"] > 0 and v["bytes"] > 0:
                    a_gbs = v["bytes"] / (v["total_ms"] * 1e-3) / 1e9
                    hk[cls] = {"kernel": label, "bound": "hbm", "achieved": a_gbs, "peak": PEAK_HBM_TBS * 1e3, "unit": "GB/s",
                               "frac": a_gbs / (PEAK_HBM_TBS * 1e3), "launches_per_step": v["launches"],
                               "avg_launch_us": v["total_ms"] * 1e3 / max(1, v["launches"]),
                               "algorithmic_bytes_per_launch": v["bytes"] / max(1, v["launches"])}
            if hk:
                # counter evidence: mean (2 FETCH_SIZE + WRITE_SIZE) x 1024 per launch of the class's
                # kernels from the committed PMC passes — those ran the TIMED configuration, i.e. per half-batch launch on a lane's
                # 128 CUs when the lanes are on, while achieved / frac above are whole-chip launches of the unsplit profile pass
                try:
                    pmc = json.load(open(os.path.join(REPO, "profiles", "latest_hbm_pmc.json")))
                    names = {"tail_self_attn": ("tail_self_attn_mfma",), "dec_cross_attn": ("tail_cross_attn_mfma", "step_cross_attn_mfma16"),
                             "enc_attn": ("enc_attn_mfma",)}
                    for cls, e in hk.items():
                        ks = [k for k in pmc["FETCH_SIZE"] if any(n in k for n in names[cls]) and k in pmc["WRITE_SIZE"]]
                        n = sum(pmc["FETCH_SIZE"][k]["launches"] for k in ks)
                        if n:
                            tot = sum((2.0 * pmc["FETCH_SIZE"][k]["sum"] + pmc["WRITE_SIZE"][k]["sum"]) * 1024.0 for k in ks)
                            e["traffic"] = tot / n
                            e["traffic_launches"] = n
                            e["traffic_source"] = ("profiles/latest_hbm_pmc.json, mean per launch over " + ", ".join(k.split("(")[0].replace("void ", "") for k in ks)
                                                   + (": launches of the timed two-lane configuration (half batches on 128 CUs each) — compare with "
                                                      "algorithmic_bytes_per_launch / 2" if lanes_on else ""))
                except Exception:
                    pass
                out["roofline_hbm_attention"] = hk
        if world == 1 and not args.no_roofline:
            try:
                out["board_power"] = board_power_under(lambda: run_step(W))
                if out["board_power"]:
                    # energy per query from the sampled mean board power and this run's rate (a change
                    # that saves watts at equal queries/s must be visible): J = W x s per step / queries per step
                    out["board_power"]["energy_j_per_query"] = out["board_power"]["power_w_mean"] * (elapsed / K) / Q
                    out["board_power"]["energy_note"] = "power_w_mean x ms_per_step / queries per step (sampled power, not an energy counter)"
            except Exception as e:
                out["board_power"] = {"error": repr(e)}
        if world == 1 and args.precision != "f32" and not args.no_exact_fp32:
            # secondary figure: the same step on the exact fp32 MFMA GEMMs (v_mfma_f32_32x32x2_f32), 1 warm-up + 5 timed
            ctx.set_precision("f32")
            run_step(W); torch.cuda.synchronize()
            n_fp32 = 5
            t0 = time.perf_counter()
            for i in range(n_fp32):
                run_step(W + (i % K))
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n_fp32
            ctx.set_precision("f16x2")
            out["exact_fp32"] = {"value": Q / dt, "unit": "queries/s", "ms_per_step": dt * 1e3, "steps": n_fp32,
                                 "dtype": "f32 (exact fp32 MFMA GEMMs)", "tail_leftover": bool(ctx.status(clear=True) & 4)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(sd, dims, B, L)
            except Exception as e:  # the baseline is reported, never required for the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "queries/s", "cores": len(os.sched_getaffinity(0)),
                                       "kind": "port", "sample": f"failed: {e!r}"}
    # ---- secondary legs: the other BASELINE configurations (see the helpers above). `train` runs on every rank (its
    # gradient all-reduce is a collective), the search legs on a single-GPU run only.
    legs = [x for x in args.secondary.split(",") if x]
    sec = {}

    def leg(name, fn):
        t0 = time.time()
        try:
            sec[name] = fn()
        except Exception as e:   # a secondary leg never takes the headline down with it
            sec[name] = {"error": repr(e)}
        if rank == 0:
            log(f"[bench] secondary {name}: {time.time() - t0:.1f}s -> "
                + json.dumps({k: v for k, v in sec[name].items() if k in ("value", "unit", "ms_per_step", "error", "len8", "beams10", "beams1000", "q1", "q8", "q64", "saturated", "batch1", "cli_batch")}))

    if "train" in legs:
        leg("train_step", lambda: secondary_train_step(E, synth, ctx, dev, world, rank, bz=args.train_bz, steps=args.train_steps))
        if ctx.has_bf16() and "train_f16x2_only" not in legs:
            leg("train_step_bf16", lambda: secondary_train_step(E, synth, ctx, dev, world, rank, bz=args.train_bz, steps=args.train_steps,
                                                                precision="bf16"))
    if world == 1:
        if "f2" in legs:
            leg("f2", lambda: secondary_f2(E, synth, ctx, model, trie, dims, dev))
        if "skew" in legs:
            leg("skew", lambda: secondary_skew(E, synth, ctx, model, dims, dev, args.docs, B, L, V, Q))
        if "v1024" in legs and args.model == "t5-base":
            leg("v1024", lambda: secondary_v1024(E, synth, ctx, dev, args.docs, B, Q))
        if "latency" in legs and args.model == "t5-base":
            leg("latency", lambda: secondary_latency(E, synth, ctx, model, trie, dims, dev, L))
        if "beam1000" in legs and args.model == "t5-base":
            leg("beam1000", lambda: secondary_beam1000(E, synth, ctx, model, trie, dims, dev, L))
        if "rankdata_ref_flags" in legs and args.model == "t5-base":
            leg("rankdata_ref_flags", lambda: secondary_rankdata_ref_flags(E, synth, ctx, model, trie, dims, dev))
        if "heavy_tail" in legs and args.model == "t5-base":
            leg("heavy_tail", lambda: secondary_heavy_tail(E, synth, ctx, trie, dims, dev, B, L, Q))
        if "small_batch" in legs and args.model == "t5-base":
            leg("small_batch", lambda: secondary_small_batch(E, synth, ctx, model, trie, dims, dev, B, L))
        if "config4" in legs and args.model == "t5-base":
            leg("config4", lambda: secondary_config4(E, synth, ctx, trie, dev, L))
        if "t5_3b" in legs and args.model == "t5-base":
            leg("t5_3b", lambda: secondary_t5_3b(E, synth, ctx, trie, dev, L))
    if rank == 0:
        out["secondary"] = sec
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
