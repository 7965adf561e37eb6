"""Secondary measurements printed inside bench.py's JSON line under "extras" (rank 0, N=1 only).

None of these is the headline `value`; they are the other BASELINE.json configurations, measured with
the same protocol (HIP-graph replay of strictly sequential launches, weights rotated through more than
the 256 MB Infinity Cache where the shape is small enough to be cached).
"""
from __future__ import annotations

import os
import time

HBM_PEAK_GBPS = 8000.0
I8_MFMA_PEAK_TOPS = 5000.0      # dense int8 MFMA peak (~2x the 2.5 PF bf16 figure)
# what an MFMA-ONLY loop (no memory, no LDS) sustains on RANDOM operands on this chip (round 3, profiles/r03_mfma_power.txt,
# tools/microbench/mfma_data_power.hip): the power budget holds the nominal dense peaks on constant operands only
F16_MFMA_RANDOM_DATA_TFLOPS = 1700.0
I8_MFMA_RANDOM_DATA_TOPS = 3270.0


def _graph_time(torch, device, fn, reps=3):
    """Capture fn() once into a HIP graph and return the best per-replay event time in ms."""
    stream = torch.cuda.Stream(device=device)
    with torch.cuda.stream(stream):
        fn()                                            # warm-up / lazy repacks outside capture
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            fn()
        g.replay()
        stream.synchronize()
        best = float("inf")
        for _ in range(reps):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            g.replay()
            e1.record(stream)
            stream.synchronize()
            best = min(best, e0.elapsed_time(e1))
    return best


def _w4_layer(torch, device, K, N, bias, gen, dtype=None):
    from chatglm_q_amd.int4.qlinear import DynamicQuantizeLinear
    dtype = dtype or torch.float16
    layer = DynamicQuantizeLinear(K, N, bias=bias, dtype=dtype, device=device)
    layer.weight.copy_(torch.randint(0, 256, layer.weight.shape, dtype=torch.uint8, device=device, generator=gen))
    layer.weight_scale.copy_((torch.rand(layer.weight_scale.shape, device=device, generator=gen) * 0.02 + 0.002).to(dtype))
    if bias:
        layer.bias.copy_((torch.randn(N, device=device, generator=gen) * 0.1).to(dtype))
    return layer.prepare()


def token_sweep(torch, device):
    """All 113 QLinear calls of one ChatGLM2-6B int4g32 decode token, M = 1, each with
    its own weights (3.36 GB streamed once per replay): the bandwidth-bound form of the hot path."""
    gen = torch.Generator(device=device).manual_seed(7)
    shapes = [(4096, 4608, True), (4096, 4096, False), (4096, 27392, False), (13696, 4096, False)]
    layers = []
    total = 0
    for _ in range(28):
        for (K, N, b) in shapes:
            layers.append(_w4_layer(torch, device, K, N, b, gen))
            total += K * N // 2 + (K // 32) * N * 2 + K * 2 + N * 2 + (N * 2 if b else 0)
    layers.append(_w4_layer(torch, device, 4096, 65024, False, gen))
    total += 4096 * 65024 // 2 + 128 * 65024 * 2 + 4096 * 2 + 65024 * 2
    xs = {4096: torch.randn(1, 4096, device=device, dtype=torch.float16),
          13696: torch.randn(1, 13696, device=device, dtype=torch.float16)}

    def fn():
        with torch.no_grad():
            for l in layers:
                l(xs[l.in_features])

    ms = _graph_time(torch, device, fn)
    gbps = total / (ms * 1e-3) / 1e9
    return {"workload": "113 int4g32 QLinear forwards of one ChatGLM2-6B decode token (M=1), linear layers only",
            "algorithmic_bytes": total, "ms": round(ms, 4), "GBps": round(gbps, 1),
            "frac_of_8TBps": round(gbps / HBM_PEAK_GBPS, 4), "linear_only_tok_per_s": round(1e3 / ms, 1)}


def _probe_lib():
    """The span / probe build of the library (csrc/probe_kernels.hip: floor probes, never part of the product library), or None."""
    import ctypes
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "chatglm_q_amd", "csrc", "libqlinear_hip_span.so")
    if not os.path.exists(path):
        return None
    lib = ctypes.CDLL(path)
    if not hasattr(lib, "qlinear_probe_read"):
        return None
    lib.qlinear_probe_read.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    lib.qlinear_probe_read_waves.restype = ctypes.c_int64
    lib.qlinear_probe_read_waves.argtypes = [ctypes.c_int64]
    return lib


def _pure_read_us(torch, device, nbytes, n_sets, reps):
    """A pure streaming read of `nbytes` per launch (probe kernel: coalesced 16-byte loads, XOR, one store per wave), regions rotated
    through n_sets x nbytes like the weight sets of the kernel it stands beside, under the same graph protocol: launch boundary + HBM
    round trip + the bytes - what a dependent launch of this size costs with no arithmetic at all.  None without the probe library."""
    lib = _probe_lib()
    if lib is None:
        return None
    region = (nbytes + 255) // 256 * 256
    buf = torch.empty(region * n_sets, dtype=torch.uint8, device=device)
    buf.random_(0, 256)
    sink = torch.zeros(int(lib.qlinear_probe_read_waves(nbytes)) + 64, dtype=torch.int32, device=device)

    def fn():
        st = torch.cuda.current_stream(device).cuda_stream
        for _ in range(reps):
            for i in range(n_sets):
                if lib.qlinear_probe_read(buf.data_ptr() + i * region, nbytes, sink.data_ptr(), st) != 0:
                    raise RuntimeError("probe launch failed")

    ms = _graph_time(torch, device, fn) / (reps * n_sets)
    del buf
    return ms * 1e3


def per_shape(torch, device):
    """Per-shape decode GEMV figures, each rotated over enough weight sets to exceed the Infinity Cache; beside each the pure
    streaming read of the same bytes under the same protocol (round 5: the floor of ONE dependent launch of that size on this box)."""
    gen = torch.Generator(device=device).manual_seed(9)
    out = {}
    for name, K, N, b in [("qkv_proj", 4096, 4608, True), ("o_proj", 4096, 4096, False), ("w_in", 4096, 27392, False),
                          ("w_out", 13696, 4096, False), ("lm_head", 4096, 65024, False)]:
        per = K * N // 2 + (K // 32) * N * 2 + K * 2 + N * 2 + (N * 2 if b else 0)
        n_sets = max(2, min(48, int(700e6 // per) + 1))
        layers = [_w4_layer(torch, device, K, N, b, gen) for _ in range(n_sets)]
        x = torch.randn(1, K, device=device, dtype=torch.float16)
        reps = max(1, 96 // n_sets)

        def fn():
            with torch.no_grad():
                for _ in range(reps):
                    for l in layers:
                        l(x)

        ms = _graph_time(torch, device, fn) / (reps * n_sets)
        out[name] = {"K": K, "N": N, "us": round(ms * 1e3, 3), "GBps": round(per / (ms * 1e-3) / 1e9, 1),
                     "frac_of_8TBps": round(per / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
        del layers
        torch.cuda.empty_cache()
        try:
            floor = _pure_read_us(torch, device, per, n_sets, reps)
        except Exception as e:      # a probe never invalidates the figure it stands beside
            floor = None
            out[name]["pure_read_error"] = repr(e)[:120]
        if floor:
            out[name]["pure_read_us"] = round(floor, 3)
            out[name]["frac_of_pure_read"] = round(floor / (ms * 1e3), 4)
        torch.cuda.empty_cache()
    # one decode token's 113 linear launches priced at the pure-read floor of their shapes (28 layers x 4 + lm_head)
    try:
        tok = 28 * sum(out[n]["pure_read_us"] for n in ("qkv_proj", "o_proj", "w_in", "w_out")) + out["lm_head"]["pure_read_us"]
        mine = 28 * sum(out[n]["us"] for n in ("qkv_proj", "o_proj", "w_in", "w_out")) + out["lm_head"]["us"]
        out["token_at_pure_read_floor"] = {"floor_us": round(tok, 1), "kernels_us": round(mine, 1), "frac": round(tok / mine, 4),
                                           "note": "113 plain GEMV launches of a token at M = 1 against pure streaming reads of the same bytes, "
                                                   "launch for launch"}
    except KeyError:
        pass
    return out


def per_shape_bf16(torch, device):
    """The same five decode shapes with bf16 activations / scales (`torch_dtype: "bfloat16"` is a valid load config of the reference,
    chatglm_q/loader.py:16-38): the module default for bf16 is the STRICT arithmetic (the reference's per-weight rounding to bf16:
    skipping it cannot land within 1e-3), which costs an fp32 product + v_cvt_pk_bf16_f32 per weight."""
    gen = torch.Generator(device=device).manual_seed(9)
    out = {}
    for name, K, N, b in [("qkv_proj", 4096, 4608, True), ("o_proj", 4096, 4096, False), ("w_in", 4096, 27392, False),
                          ("w_out", 13696, 4096, False), ("lm_head", 4096, 65024, False)]:
        per = K * N // 2 + (K // 32) * N * 2 + K * 2 + N * 2 + (N * 2 if b else 0)
        n_sets = max(2, min(48, int(700e6 // per) + 1))
        reps = max(1, 96 // n_sets)
        row = {}
        for label, dt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
            layers = [_w4_layer(torch, device, K, N, b, gen, dt) for _ in range(n_sets)]
            x = torch.randn(1, K, device=device, dtype=dt)

            def fn():
                with torch.no_grad():
                    for _ in range(reps):
                        for l in layers:
                            l(x)

            row[label + "_us"] = round(_graph_time(torch, device, fn) / (reps * n_sets) * 1e3, 3)
            del layers
            torch.cuda.empty_cache()
        row["bf16_over_f16"] = round(row["bf16_us"] / row["f16_us"], 4)
        out[name] = row
    return out


def e2e_generate_bf16(torch, device):
    """BASELINE config 4's harness with a bf16 model (strict arithmetic: the module default for bf16): greedy and the reference's default
    sampler, graph-replayed, 128 tokens."""
    from chatglm_q_amd.decoder import ChatGLMDecoder
    model, cfg = _chatglm2_6b(torch, device, torch.bfloat16)
    for mod in model.modules():
        if hasattr(mod, "prepare"):
            mod.prepare()
    prompt = [(37 * i + 11) % cfg.vocab_size for i in range(32)]
    dec = ChatGLMDecoder(None, model)
    out = {}
    for kw in (dict(greedy=True), dict(seed=1)):
        list(dec.generate_ids(prompt, max_generated_tokens=8, ignore_eos=True, use_graph=True, **kw))
    for label, kw in (("greedy", dict(greedy=True)), ("sampled_default", dict(seed=20260930))):
        toks = list(dec.generate_ids(prompt, max_generated_tokens=128, ignore_eos=True, use_graph=True, **kw))
        out[label] = {"generated": len(toks), "gen_tok_per_s": round(dec.last_stats["gen_tok_per_s"], 1),
                      "avg_tok_per_s": round(dec.last_stats["avg_tok_per_s"], 1)}
    out["workload"] = "ChatGLM2-6B int4g32 generate(), batch 1, 32-token prompt, 128 generated tokens, bf16 (strict per-weight rounding), synthetic weights"
    del model
    torch.cuda.empty_cache()
    return out


def e2e_generate_int8(torch, device):
    """The reference's other model family (`quant_type: "int8"`, chatglm_q/loader.py:16-38) through the same harness: per-channel int8
    weights, weight-only, fp16 activations - twice the int4g32 model's bytes per token."""
    from chatglm_q_amd import model as M
    from chatglm_q_amd.decoder import ChatGLMDecoder
    cfg = M.ChatGLM2Config()
    with torch.device(device):
        model = M.create_quant_int8_model(cfg, dtype=torch.float16)
    M.fill_synthetic_(model, 0)
    model.eval()
    prompt = [(37 * i + 11) % cfg.vocab_size for i in range(32)]
    dec = ChatGLMDecoder(None, model)
    out = {}
    for kw in (dict(greedy=True), dict(seed=1)):
        list(dec.generate_ids(prompt, max_generated_tokens=8, ignore_eos=True, use_graph=True, **kw))
    for label, kw in (("greedy", dict(greedy=True)), ("sampled_default", dict(seed=20260930))):
        toks = list(dec.generate_ids(prompt, max_generated_tokens=128, ignore_eos=True, use_graph=True, **kw))
        out[label] = {"generated": len(toks), "gen_tok_per_s": round(dec.last_stats["gen_tok_per_s"], 1),
                      "avg_tok_per_s": round(dec.last_stats["avg_tok_per_s"], 1)}
    bytes_per_token = 2 * LINEAR_BYTES_PER_TOKEN * 0.94      # int8 codes: 2 x the nibbles; scales per channel instead of per group
    out["greedy"]["linear_frac_of_8TBps"] = round(out["greedy"]["gen_tok_per_s"] * bytes_per_token / 8e12, 4)
    out["workload"] = "ChatGLM2-6B int8 per-channel generate(), batch 1, 32-token prompt, 128 generated tokens, fp16, synthetic weights"
    del model
    torch.cuda.empty_cache()
    return out


def w8_decode(torch, device):
    from chatglm_q_amd.int8.qlinear import DynamicQuantizeLinear
    gen = torch.Generator(device=device).manual_seed(11)
    K = N = 4096
    per = K * N + N * 2 + K * 2 + N * 2
    layers = []
    for _ in range(40):
        l = DynamicQuantizeLinear(K, N, bias=False, dtype=torch.float16, device=device)
        l.weight.copy_(torch.randint(-127, 128, l.weight.shape, dtype=torch.int8, device=device, generator=gen))
        l.weight_scale.copy_((torch.rand(N, device=device, generator=gen) * 0.01 + 0.001).half())
        layers.append(l)
    x = torch.randn(1, K, device=device, dtype=torch.float16)

    def fn():
        with torch.no_grad():
            for _ in range(3):
                for l in layers:
                    l(x)

    ms = _graph_time(torch, device, fn) / (3 * len(layers))
    return {"workload": "int8 per-channel QLinear 1x4096->4096 fp16 (weight-only)", "us": round(ms * 1e3, 3),
            "GBps": round(per / (ms * 1e-3) / 1e9, 1), "frac_of_8TBps": round(per / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}


def _c3_rocprof():
    """Kernel averages of BASELINE config 3 under rocprofv3 from the newest tracked summary (profiles/rNN_summary.json,
    w8a8_config3_under_rocprofv3: tools/profile_round.sh + tools/profile_w8a8_c3.py)."""
    import glob
    import json
    root = os.path.dirname(os.path.abspath(__file__))
    for f in reversed(sorted(glob.glob(os.path.join(root, "profiles", "r*_summary.json")))):
        try:
            d = json.load(open(f))["w8a8_config3_under_rocprofv3"]
            d["source"] = os.path.relpath(f, root)
            return d
        except (KeyError, ValueError, OSError):
            continue
    return None


def w8a8_config3(torch, device):
    """BASELINE config 3: int8 per-channel weights, int8-quantised activations, 512x4096->4096, through the module path
    (tile-major weights, qlinear_w8a8_linear_tiled); the two kernels also timed apart, and round 1's row-major path."""
    from chatglm_q_amd.int8 import hip_ops
    gen = torch.Generator(device=device).manual_seed(13)
    M, K, N = 512, 4096, 4096
    ws = [torch.randint(-127, 128, (N, K), dtype=torch.int8, device=device, generator=gen) for _ in range(20)]
    tiled = [hip_ops.tile_w8(w) for w in ws]
    sc = (torch.rand(N, device=device, generator=gen) * 0.01 + 0.001).half()
    a = torch.randn(M, K, device=device, dtype=torch.float16)
    a_q, a_s = hip_ops.act_quant_rowwise(a)
    n = len(ws)
    us = _graph_time(torch, device, lambda: [hip_ops.w8a8_forward_tiled(a, t, N, sc) for t in tiled]) / n * 1e3
    us_q = _graph_time(torch, device, lambda: [hip_ops.act_quant_rowwise(a) for _ in range(n)]) / n * 1e3
    us_g = _graph_time(torch, device, lambda: [hip_ops.w8a8_gemm_tiled(a_q, a_s, t, N, sc) for t in tiled]) / n * 1e3
    us_r1 = _graph_time(torch, device, lambda: [hip_ops.w8a8_forward(a, w, sc) for w in ws]) / n * 1e3
    # in a model the rows come out of a norm / activation launch: with the quantising producer (qlinear_rmsnorm_quant_i8) the int8
    # rows + scales are a by-product of THAT launch - what the int8-activation linear itself costs is then the GEMM launch
    from chatglm_q_amd import fused_ops as F_
    lnw = torch.ones(K, device=device, dtype=torch.float16)
    us_norm = _graph_time(torch, device, lambda: [F_.rmsnorm(a, lnw, 1e-5) for _ in range(n)]) / n * 1e3
    us_normq = _graph_time(torch, device, lambda: [F_.rmsnorm_quant(a, lnw, 1e-5)[1] for _ in range(n)]) / n * 1e3
    us_chain_sep = _graph_time(torch, device, lambda: [hip_ops.w8a8_forward_tiled(F_.rmsnorm(a, lnw, 1e-5), t, N, sc) for t in tiled]) / n * 1e3

    def chain_fused():
        for t in tiled:
            _, q, s_, _ = F_.rmsnorm_quant(a, lnw, 1e-5)
            hip_ops.w8a8_gemm_tiled(q, s_, t, N, sc)
    us_chain_fused = _graph_time(torch, device, chain_fused) / n * 1e3
    ops = 2.0 * M * N * K
    # yardstick, never the target: the vendor library's plain i8 x i8 -> i32 GEMM (torch._int_mm -> hipBLASLt) on the same box,
    # no scales, no output conversion
    yard = {}
    for Mv in (512, 8192):
        try:
            av = torch.randint(-127, 128, (Mv, K), dtype=torch.int8, device=device, generator=gen)
            wts = [w.t() for w in ws[:8]]
            usv = _graph_time(torch, device, lambda: [torch._int_mm(av, wt) for wt in wts]) / len(wts) * 1e3
            yard[f"M{Mv}"] = {"us": round(usv, 2), "TOPs": round(2.0 * Mv * N * K / usv / 1e6, 1)}
        except Exception as e:      # not every build ships the op
            yard[f"M{Mv}"] = {"error": repr(e)[:200]}
    prof = _c3_rocprof()
    return {"workload": "act-quant + i8xi8 MFMA GEMM 512x4096->4096 (fused op time, both kernels), tile-major weights",
            "vendor_i8_gemm_yardstick_torch_int_mm": yard,
            # the same fraction from the tracked rocprofv3 kernel averages of this workload alone (GEMM + quantiser)
            "frac_rocprof": (round(ops / ((prof["gemm_avg_ns"] + prof["act_quant_avg_ns"]) * 1e-9) / 1e12 / I8_MFMA_PEAK_TOPS, 4) if prof else None),
            "gemm_frac_rocprof": (round(ops / (prof["gemm_avg_ns"] * 1e-9) / 1e12 / I8_MFMA_PEAK_TOPS, 4) if prof else None),
            "rocprof_source": prof.get("source") if prof else None,
            "us": round(us, 2), "TOPs": round(ops / us / 1e6, 1),
            "frac_of_i8_mfma_peak": round(ops / us / 1e6 / I8_MFMA_PEAK_TOPS, 4),
            "act_quant_us": round(us_q, 2), "gemm_us": round(us_g, 2), "gemm_TOPs": round(ops / us_g / 1e6, 1),
            "row_major_weights_round1_path_us": round(us_r1, 2),
            "with_quantising_producer": {
                "note": "RMSNorm -> int8-activation linear: 3 launches (norm, quantiser, GEMM) vs 2 (norm emitting int8 rows + scales, GEMM)",
                "rmsnorm_us": round(us_norm, 2), "rmsnorm_quant_us": round(us_normq, 2),
                "norm_quantiser_gemm_us": round(us_chain_sep, 2), "norm_quant_gemm_us": round(us_chain_fused, 2),
                "linear_cost_us": round(us_chain_fused - us_norm, 2),
                "linear_TOPs": round(ops / max(us_chain_fused - us_norm, 1e-3) / 1e6, 1),
                "linear_frac_of_i8_mfma_peak": round(ops / max(us_chain_fused - us_norm, 1e-3) / 1e6 / I8_MFMA_PEAK_TOPS, 4)}}


def prefill_gemm(torch, device):
    """BASELINE config 5 building block: int4g32 GEMM with M = 8192 rows (seq 2048 x batch 4) per layer shape,
    fp16 MFMA with in-register dequant (reference rounding sequence)."""
    gen = torch.Generator(device=device).manual_seed(17)
    M = int(os.environ.get("PREFILL_GEMM_M", 8192))          # tools/prefill_gemm_ab.py sweeps it
    out = {}
    for name, K, N in [("qkv_proj", 4096, 4608), ("o_proj", 4096, 4096), ("w_in", 4096, 27392), ("w_out", 13696, 4096),
                       ("lm_head", 4096, 65024)]:          # every layer shape (a real prefill runs lm_head on the last position only)
        # four weight sets per captured graph: a fresh set of weights per launch (as in a model), and the ~10 us a graph replay costs on
        # its own (MI355X_MICROARCH.md, graph-replay-floor) is spread over four launches instead of charged to one 250 us kernel
        layers = [_w4_layer(torch, device, K, N, False, gen) for _ in range(4)]
        x = torch.randn(M, K, device=device, dtype=torch.float16)

        def fn():
            with torch.no_grad():
                for layer in layers:
                    layer(x)

        ms = _graph_time(torch, device, fn) / len(layers)
        flops = 2.0 * M * N * K
        # yardstick, never the target: the vendor library's DENSE f16 GEMM (hipBLASLt through torch) on the same box, same protocol -
        # it streams 4 x the weight bytes and dequantises nothing
        try:
            wd = [torch.randn(N, K, device=device, dtype=torch.float16) * 0.05 for _ in range(4)]
            ms_v = _graph_time(torch, device, lambda: [x @ w.t() for w in wd]) / len(wd)
            vendor = round(flops / (ms_v * 1e-3) / 1e12, 1)
            del wd
        except Exception as e:      # pragma: no cover
            vendor = repr(e)[:120]
        out[name] = {"M": M, "K": K, "N": N, "ms": round(ms, 4), "TFLOPs": round(flops / (ms * 1e-3) / 1e12, 1),
                     "vendor_dense_f16_gemm_yardstick_TFLOPs": vendor,
                     "frac_of_2.5PF_f16_mfma": round(flops / (ms * 1e-3) / 1e12 / 2500.0, 4),
                     "frac_of_mfma_only_loop_on_random_data_1.7PF": round(flops / (ms * 1e-3) / 1e12 / F16_MFMA_RANDOM_DATA_TFLOPS, 4)}
        del layers, x
        torch.cuda.empty_cache()
    return out


def _smi_sampler(samples, stop):
    """Background thread body: socket power (W) and shader clock (MHz) from rocm-smi, ~5 samples per second."""
    import re
    import subprocess
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5).stdout
            rows = [r for r in txt.splitlines() if r.startswith("card")]
            head = [r for r in txt.splitlines() if r.startswith("device")]
            if rows and head:
                cols, vals = head[0].split(","), rows[0].split(",")
                rec = {}
                for c, v in zip(cols, vals):
                    if c.startswith("sclk clock speed"):
                        m = re.search(r"(\d+)", v)
                        rec["sclk_mhz"] = int(m.group(1)) if m else None
                    if "Power" in c:
                        try:
                            rec["watts"] = float(v)
                        except ValueError:
                            pass
                if rec:
                    samples.append(rec)
        except Exception:      # noqa: BLE001 - no rocm-smi on the box: the probe reports nothing
            return
        stop.wait(0.15)


def prefill_gemm_power(torch, device):
    """What bounds BASELINE config 5's GEMMs (round 5): the o_proj-shaped int4g32 GEMM (8192 x 4096 x 4096) and the vendor's dense f16 GEMM
    launched back to back for ~2.5 s each while rocm-smi samples socket power and the shader clock.  Round 5's finding
    (profiles/r05_g256_power_cap.txt): the kernel holds the chip at its 1 400 W cap with the shader clock at ~1.7 GHz of 2.4; on 64
    of the 256 CUs the same kernel runs at 2.4 GHz and delivers 35 % of the full chip's throughput with 25 % of its CUs."""
    import threading
    gen = torch.Generator(device=device).manual_seed(23)
    M, K, N = 8192, 4096, 4096
    layers = [_w4_layer(torch, device, K, N, False, gen) for _ in range(4)]
    x = torch.randn(M, K, device=device, dtype=torch.float16)
    wd = [torch.randn(N, K, device=device, dtype=torch.float16) * 0.05 for _ in range(4)]
    out = {}
    for name, body in [("int4g32_gemm256", lambda: [l(x) for l in layers]), ("vendor_dense_f16", lambda: [x @ w.t() for w in wd])]:
        samples, stop = [], threading.Event()
        th = threading.Thread(target=_smi_sampler, args=(samples, stop))
        with torch.no_grad():
            body()
            torch.cuda.synchronize()
            th.start()
            t0, n = time.perf_counter(), 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            while time.perf_counter() - t0 < 2.5:
                for _ in range(25):
                    body()
                n += 100
                torch.cuda.synchronize()
            e1.record()
            torch.cuda.synchronize()
        stop.set()
        th.join()
        us = e0.elapsed_time(e1) * 1e3 / n
        mid = samples[len(samples) // 4:] or samples          # drop the ramp
        watts = sorted(r["watts"] for r in mid if r.get("watts") is not None)
        sclk = sorted(r["sclk_mhz"] for r in mid if r.get("sclk_mhz") is not None)
        out[name] = {"us_per_launch_sustained": round(us, 1), "TFLOPs_sustained": round(2.0 * M * N * K / us / 1e6, 1), "launches": n,
                     "socket_watts_median": watts[len(watts) // 2] if watts else None,
                     "sclk_mhz_median": sclk[len(sclk) // 2] if sclk else None, "smi_samples": len(mid)}
    out["note"] = ("sustained back-to-back launches (host loop, not the graph protocol of prefill_gemm_M8192); the board power cap is 1 400 W and "
                   "the nominal shader clock 2 400 MHz: a kernel that sits at the cap with the clock pulled down is bound by energy per flop")
    return out


def fp32_many_rows(torch, device):
    """fp32 activations at 512 x 4096 -> 4096 (round 5, VERDICT r4 missing 2): the fp32 matrix-instruction kernel (wq_gemm_f32.hip,
    v_mfma_f32_32x32x2_f32 on the canonical buffers, exact fp32 products and sums) beside the VALU kernels that served those calls before
    (QLINEAR_DISPATCH=nof32mfma), int4g32 and int8 per channel; 157 TFLOP/s = the chip's dense fp32 matrix peak."""
    from chatglm_q_amd import _lib
    from chatglm_q_amd.int4 import hip_ops as h4
    from chatglm_q_amd.int8 import hip_ops as h8
    lib = _lib.get_lib()
    gen = torch.Generator(device=device).manual_seed(29)
    M, K, N = 512, 4096, 4096
    a = torch.randn(M, K, device=device, generator=gen)
    qw = torch.randint(0, 256, (K // 2, N), dtype=torch.uint8, device=device, generator=gen)
    sc = torch.rand(K // 32, N, device=device, generator=gen) * 0.02 + 0.002
    w8 = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=device, generator=gen)
    s8 = torch.rand(N, device=device, generator=gen) * 0.01 + 0.001
    out = {"M": M, "K": K, "N": N}
    prev = os.environ.get("QLINEAR_DISPATCH")
    try:
        for tag, env in (("matrix_kernel", None), ("valu_kernels", "nof32mfma")):
            if env:
                os.environ["QLINEAR_DISPATCH"] = env
            else:
                os.environ.pop("QLINEAR_DISPATCH", None)
            lib.qlinear_dispatch_reload()
            us4 = _graph_time(torch, device, lambda: h4.w4_forward(a, qw, sc)) * 1e3
            us8 = _graph_time(torch, device, lambda: h8.w8_forward(a, w8.t(), s8)) * 1e3
            out[tag] = {"int4g32_us": round(us4, 1), "int4g32_TFLOPs": round(2.0 * M * N * K / us4 / 1e6, 1),
                        "int8_us": round(us8, 1), "int8_TFLOPs": round(2.0 * M * N * K / us8 / 1e6, 1)}
    finally:
        if prev is None:
            os.environ.pop("QLINEAR_DISPATCH", None)
        else:
            os.environ["QLINEAR_DISPATCH"] = prev
        lib.qlinear_dispatch_reload()
    out["speedup_int4g32"] = round(out["valu_kernels"]["int4g32_us"] / out["matrix_kernel"]["int4g32_us"], 2)
    out["speedup_int8"] = round(out["valu_kernels"]["int8_us"] / out["matrix_kernel"]["int8_us"], 2)
    out["frac_of_157TF_fp32_matrix_peak_int4g32"] = round(out["matrix_kernel"]["int4g32_TFLOPs"] / 157.3, 4)
    return out


def int8_prefill_gemm(torch, device):
    """The int8 model's many-row GEMMs at M = 8192 (o_proj and w_out shapes): weight-only (the reference's int8 forward,
    chatglm_q/int8/triton_ops.py:62-73) in TFLOP/s and int8-activation (act_quant) in TOP/s, GEMM launch alone."""
    from chatglm_q_amd.int8 import hip_ops
    gen = torch.Generator(device=device).manual_seed(19)
    M, out = 8192, {}
    for name, K, N in [("o_proj", 4096, 4096), ("w_out", 13696 - 13696 % 128, 4096)]:
        tiled = [hip_ops.tile_w8(torch.randint(-127, 128, (N, K), dtype=torch.int8, device=device, generator=gen)) for _ in range(3)]
        sc = (torch.rand(N, device=device, generator=gen) * 0.01 + 0.001).half()
        a = torch.randn(M, K, device=device, dtype=torch.float16)
        a_q, a_s = hip_ops.act_quant_rowwise(a)
        us_w = _graph_time(torch, device, lambda: [hip_ops.w8_forward_tiled(a, t, N, sc) for t in tiled]) / len(tiled) * 1e3
        us_a = _graph_time(torch, device, lambda: [hip_ops.w8a8_gemm_tiled(a_q, a_s, t, N, sc) for t in tiled]) / len(tiled) * 1e3
        ops = 2.0 * M * N * K
        out[name] = {"M": M, "K": K, "N": N, "weight_only_us": round(us_w, 1), "weight_only_TFLOPs": round(ops / us_w / 1e6, 1),
                     "weight_only_frac_of_2.5PF_f16_mfma": round(ops / us_w / 1e6 / 2500.0, 4),
                     "weight_only_frac_of_mfma_only_loop_on_random_data_1.7PF": round(ops / us_w / 1e6 / F16_MFMA_RANDOM_DATA_TFLOPS, 4),
                     "int8_activations_frac_of_mfma_only_loop_on_random_data_3.27POPs": round(ops / us_a / 1e6 / I8_MFMA_RANDOM_DATA_TOPS, 4),
                     "int8_activations_us": round(us_a, 1), "int8_activations_TOPs": round(ops / us_a / 1e6, 1),
                     "int8_activations_frac_of_i8_mfma_peak": round(ops / us_a / 1e6 / I8_MFMA_PEAK_TOPS, 4)}
        # the same GEMM on UNIFORM random int8 rows (quantised Gaussian rows toggle fewer bits: these GEMMs are power bound, the
        # operand data moves the result by ~3 %) beside the vendor's plain i8 GEMM on exactly those operands
        try:
            au = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=device, generator=gen)
            us_u = _graph_time(torch, device, lambda: [hip_ops.w8a8_gemm_tiled(au, a_s, t, N, sc) for t in tiled]) / len(tiled) * 1e3
            wv = [torch.randint(-127, 128, (N, K), dtype=torch.int8, device=device, generator=gen).t() for _ in range(3)]
            us_v = _graph_time(torch, device, lambda: [torch._int_mm(au, w) for w in wv]) / len(wv) * 1e3
            out[name]["uniform_random_operands"] = {"int8_activations_TOPs": round(ops / us_u / 1e6, 1),
                                                    "vendor_i8_gemm_yardstick_torch_int_mm_TOPs": round(ops / us_v / 1e6, 1)}
            del wv, au
        except Exception as e:      # pragma: no cover
            out[name]["uniform_random_operands"] = {"error": repr(e)[:120]}
        del tiled
        torch.cuda.empty_cache()
    return out


def _chatglm2_6b(torch, device, dtype, seed=0):
    from chatglm_q_amd import model as M
    cfg = M.ChatGLM2Config()
    prev = torch.get_default_device() if hasattr(torch, "get_default_device") else None
    with torch.device(device):
        m = M.create_quant_int4_model(cfg, dtype=dtype)
    M.fill_synthetic_(m, seed)
    return m.eval(), cfg


LINEAR_BYTES_PER_TOKEN = 3362254848       # QLinear bytes touched per decoded token


def drop_in_generate(torch, model, prompt, n_tokens=24):
    """What a maintainer gets by binding ONLY the two QLinear entry points and keeping the
    reference's own graph and loop: every module call is one eager launch through ctypes, everything around it is plain
    torch in the reference's op order (no fused decode ops, no static cache - the cache grows by torch.cat -, lm_head over
    every position, no HIP graph), one host sync per token as in chatglm_q/decoder.py:76-91."""
    from chatglm_q_amd import model as M
    prev = M.FUSED_DECODE_OPS
    M.FUSED_DECODE_OPS = False
    try:
        ids = torch.tensor([prompt], device=model.final_ln.weight.device)
        times, kv = [], None
        with torch.no_grad():
            for _ in range(n_tokens):
                t0 = time.perf_counter()
                _, logits, kv = model(input_ids=ids, past_key_values=kv)
                tok = int(logits[0, -1].argmax().item())
                times.append(time.perf_counter() - t0)
                ids = torch.tensor([[tok]], device=ids.device)
        rest = sum(times[1:])
        return {"generated": n_tokens, "prefill_s": round(times[0], 4), "gen_tok_per_s": round((n_tokens - 1) / rest, 1),
                "note": "reference-shaped forward(): eager QLinear launches + plain torch ops, cache grown by torch.cat"}
    finally:
        M.FUSED_DECODE_OPS = prev


def _fused_vs_plain(torch, model, prompt):
    """rel-L2 between the last-position logits of the fused path (prefill + 4 graph-replayed steps) and of the reference-shaped
    forward() with the fused decode ops switched off, same model, same tokens."""
    from chatglm_q_amd import model as M
    from chatglm_q_amd.decoder import DecodeSession
    dev = model.final_ln.weight.device
    ids = torch.tensor([prompt], device=dev)
    sess = DecodeSession(model, 1, 64, use_graph=True)
    lg = sess.prefill(ids)
    toks = []
    sess.tok.copy_(lg.argmax(-1, keepdim=True))
    sess.capture(greedy=True)
    for _ in range(4):
        toks.append(int(sess.tok.item()))
        lg = sess.decode_step(greedy=True)
    fused = lg.float()
    prev = M.FUSED_DECODE_OPS
    M.FUSED_DECODE_OPS = False
    try:
        with torch.no_grad():
            _, logits, _ = model(input_ids=torch.tensor([prompt + toks], device=dev))
    finally:
        M.FUSED_DECODE_OPS = prev
    plain = logits[:, -1].float()
    return round(float((fused - plain).norm() / plain.norm()), 6)


def _resident_bytes(model):
    """Bytes a decode + prefill deployment of the model holds: canonical buffers and every derived layout built so far."""
    tot = {}
    for m in model.modules():
        if hasattr(m, "derived_nbytes"):
            for k, v in m.derived_nbytes().items():
                tot[k] = tot.get(k, 0) + v
    tot["total_GB"] = round(sum(v for k, v in tot.items()) / 1e9, 2)
    return tot


def e2e_generate(torch, device):
    """BASELINE config 4: ChatGLM2-6B int4g32 (synthetic weights), batch 1, prompt 32 ids, 128 generated tokens,
    greedy, EOS ignored.  Timing definitions of the reference (chatglm_q/decoder.py:99-106)."""
    from chatglm_q_amd.decoder import ChatGLMDecoder
    model, cfg = _chatglm2_6b(torch, device, torch.float16)
    for mod in model.modules():
        if hasattr(mod, "prepare"):
            mod.prepare()
    prompt = [(37 * i + 11) % cfg.vocab_size for i in range(32)]
    dec = ChatGLMDecoder(None, model)
    out = {}
    # untimed warm-up of every mode (lazy layouts, graph capture, allocator): round 2 timed them inside the first leg
    for kw in (dict(use_graph=True, sync_every_token=True), dict(use_graph=False)):
        list(dec.generate_ids(prompt, max_generated_tokens=8, greedy=True, ignore_eos=True, **kw))
    streams = {}
    for label, kw in [("graph_sync_every_token", dict(use_graph=True, sync_every_token=True)),
                      ("graph_device_loop", dict(use_graph=True, sync_every_token=False)),
                      ("eager", dict(use_graph=False))]:
        n = 128 if label != "eager" else 32
        toks = list(dec.generate_ids(prompt, max_generated_tokens=n, greedy=True, ignore_eos=True, **kw))
        streams[label] = toks
        s = dec.last_stats
        out[label] = {"generated": len(toks), "prefill_s": round(s["init_s"], 4),
                      "gen_tok_per_s": round(s["gen_tok_per_s"], 1), "avg_tok_per_s": round(s["avg_tok_per_s"], 1),
                      "linear_GBps": round(s["gen_tok_per_s"] * LINEAR_BYTES_PER_TOKEN / 1e9, 1),
                      "linear_frac_of_8TBps": round(s["gen_tok_per_s"] * LINEAR_BYTES_PER_TOKEN / 8e12, 4)}
    # The reference's ONLY decoding mode (chatglm_q/decoder.py:85: top_p_sampling(top_k 100, top_p 0.8, T 1.0) per token) = the default call of
    # generate_ids: "sampled_default" runs it as one launch inside the captured step (csrc/sampler.hip, round 6), "sampled_host_torch" as round 5
    # ran it - the composed torch ops (a 65 024-wide sort among them) on the host's side of the loop with an H2D copy of the token per step.
    for kw in (dict(use_graph=True, seed=1), dict(use_graph=True, device_sampler=False)):
        list(dec.generate_ids(prompt, max_generated_tokens=8, ignore_eos=True, **kw))
    for label, kw in [("sampled_default", dict(use_graph=True, seed=20260930)),
                      ("sampled_default_device_loop", dict(use_graph=True, seed=20260930, sync_every_token=False)),
                      ("sampled_host_torch", dict(use_graph=True, device_sampler=False))]:
        toks = list(dec.generate_ids(prompt, max_generated_tokens=128, ignore_eos=True, **kw))      # greedy=False: the default
        streams[label] = toks
        s = dec.last_stats
        out[label] = {"generated": len(toks), "gen_tok_per_s": round(s["gen_tok_per_s"], 1), "avg_tok_per_s": round(s["avg_tok_per_s"], 1),
                      "distinct_tokens": len(set(toks))}
    out["sampled_default"]["vs_greedy"] = round(out["sampled_default"]["gen_tok_per_s"] / out["graph_sync_every_token"]["gen_tok_per_s"], 4)
    out["sampled_default"]["same_tokens_in_device_loop"] = streams["sampled_default"] == streams["sampled_default_device_loop"]
    # correctness of what was just timed: the three modes emit one token stream, and the logits of a fused
    # graph-replayed step agree with the same model run through plain torch ops around eager QLinear launches
    out["check"] = {"graph_equals_device_loop": streams["graph_sync_every_token"] == streams["graph_device_loop"],
                    "graph_equals_eager_first_32": streams["graph_sync_every_token"][:32] == streams["eager"][:32],
                    "fused_graph_step_vs_plain_torch_graph_rel_l2": _fused_vs_plain(torch, model, prompt)}
    out["resident_bytes"] = _resident_bytes(model)
    out["drop_in_reference_graph"] = drop_in_generate(torch, model, prompt)
    out["workload"] = ("ChatGLM2-6B int4g32 generate(), batch 1, 32-token prompt, 128 generated tokens, fp16, synthetic weights; greedy legs + the "
                       "reference's default sampler (sampled_*)")
    # chunked prefill, BASELINE config 5: seq 2048 x batch 4, chunks of 2048 positions (M = 8192 rows per forward: every projection on
    # the 256 x 256-tile GEMM; attention = one launch per layer, csrc/prefill_attention.hip; tools/prefill_chunks.py, round 3:
    # 0.127 / 0.107 / 0.095 / 0.092 s for chunks of 256 / 512 / 1024 / 2048 - with the GEMM-route attention 0.153 / 0.131 / 0.122 / 0.130)
    from chatglm_q_amd.decoder import DecodeSession
    B, S, CH = 4, 2048, int(os.environ.get("PREFILL_CHUNK", 2048))
    ids = torch.randint(0, cfg.vocab_size, (B, S), device=device)
    lin_flops = 2.0 * B * S * (4096 * 4608 + 4096 * 4096 + 4096 * 27392 + 13696 * 4096) * 28
    sess = DecodeSession(model, B, S, use_graph=False)
    sess.prefill(ids[:, :CH], CH)                               # warm-up (lazy layouts, allocator)
    torch.cuda.synchronize()
    sess = DecodeSession(model, B, S, use_graph=False)
    t0 = time.perf_counter()
    sess.prefill(ids, CH)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["chunked_prefill_seq2048_batch4"] = {
        "chunk": CH, "seconds": round(dt, 4), "tokens_per_s": round(B * S / dt, 1),
        "linear_TFLOPs_if_all_time_were_linear": round(lin_flops / dt / 1e12, 1),
        "note": "whole forward incl. attention (qlinear_prefill_attention: one launch per layer on the matrix cores, scores never "
                "written; chatglm_q_amd.model.PREFILL_ATTENTION = False = the reference op sequence as two batched GEMMs + one mask / softmax launch, "
                "0.122 s) and norms; lm_head for the last position only"}
    # batched greedy decode (graph-replayed step, 32-token prompts): aggregate tokens per second
    bd = {}
    for Bd in (2, 4, 8, 32):
        try:
            idsb = torch.randint(0, cfg.vocab_size, (Bd, 32), device=device)
            sb = DecodeSession(model, Bd, 128, use_graph=True)
            lg = sb.prefill(idsb)
            sb.tok.copy_(lg.argmax(-1, keepdim=True))
            sb.capture(greedy=True)
            sb.decode_step(greedy=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(48):
                sb.decode_step(greedy=True)
            torch.cuda.synchronize()
            dtb = time.perf_counter() - t0
            bd[f"batch_{Bd}"] = {"ms_per_step": round(dtb / 48 * 1e3, 3), "tok_per_s": round(Bd * 48 / dtb, 1)}
            del sb
        except Exception as e:
            bd[f"batch_{Bd}"] = {"error": repr(e)}
    out["batched_decode"] = bd
    # low-footprint mode (round 5): the canonical GPU buffers dropped once the derived layouts exist (state_dict / checkpoints are
    # rebuilt byte for byte from part 1).  Measured on the model as this function leaves it (prefill + decode layouts resident), then
    # in a decode-only session; the decode speed of the low-footprint graph beside the default one.
    try:
        low = {}
        sess = DecodeSession(model, 1, 256, use_graph=True, low_footprint=True)
        idl = torch.randint(0, cfg.vocab_size, (1, 32), device=device)
        lg = sess.prefill(idl)
        sess.tok.copy_(lg.argmax(-1, keepdim=True))
        sess.capture(greedy=True)
        low["prefill_and_decode_layouts"] = _resident_bytes(model)
        del sess
        sess = DecodeSession(model, 1, 256, use_graph=True, low_footprint=True, decode_only=True)
        lg = sess.prefill(idl)
        sess.tok.copy_(lg.argmax(-1, keepdim=True))
        sess.capture(greedy=True)
        sess.decode_step(greedy=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(128):
            sess.decode_step(greedy=True)
        torch.cuda.synchronize()
        low["decode_only"] = _resident_bytes(model)
        low["decode_only_tok_per_s"] = round(128 / (time.perf_counter() - t0), 1)
        low["note"] = ("resident bytes of the quantized linear layers (canonical + derived layouts); the reference's guidance for int4g32 "
                       "is 6G+ of VRAM (readme.md:72); the default keeps the canonical copy beside the layouts")
        out["low_footprint"] = low
        del sess
    except Exception as e:      # noqa: BLE001
        out["low_footprint"] = {"error": repr(e)}
    del model
    torch.cuda.empty_cache()
    return out


def e2e_cpu(torch, device):
    """Same harness on the host CPU (reference-formula CPU branch of the modules), bounded: 4-token prompt, 2 decode steps."""
    from chatglm_q_amd.decoder import ChatGLMDecoder
    model, cfg = _chatglm2_6b(torch, torch.device("cpu"), torch.float32)
    dec = ChatGLMDecoder(None, model)
    toks = list(dec.generate_ids([11], max_generated_tokens=2, greedy=True, ignore_eos=True, use_graph=False))
    s = dec.last_stats
    return {"workload": "same model and loop on the host CPU (reference-formula CPU branch), fp32, 1-token prompt + 2 tokens",
            "threads": torch.get_num_threads(),
            "generated": len(toks), "prefill_s": round(s["init_s"], 2), "gen_tok_per_s": round(s["gen_tok_per_s"], 4)}


def decode_attention(torch, device):
    """The decode attention launch alone (ChatGLM2 geometry: 32 heads, 2 key / value groups, D = 128, batch 1; rotary + cache write + attention
    in one launch, window split + combine launch above 256 positions): 28 rotating caches in one HIP graph, per capacity, the context 20
    positions short of it; plus the same launch followed by o_proj with the prefetch workgroups, as the decode loop runs it."""
    from chatglm_q_amd import _lib, fused_ops as F_
    from chatglm_q_amd import model as M
    B, H, G, D, L = 1, 32, 2, 128, 28
    out = {}
    gen = torch.Generator(device=device).manual_seed(3)
    for cap in (256, 1152, 4224):
        n = cap - 20
        qkv = torch.randn(B, 1, (H + 2 * G) * D, device=device, generator=gen).half()
        table = M.rotary_table(D, cap + 8).to(device).half().reshape(cap + 8, -1).contiguous()
        pos = torch.full((B, 1), n + 1, dtype=torch.long, device=device)
        widx = torch.tensor([n], dtype=torch.long, device=device)
        mask = torch.full((B, 1, cap), -1e10, device=device)
        mask[:, :, : n + 1] = 0
        caches = [(torch.randn(B, cap, G, D, device=device, generator=gen).half(), torch.randn(B, cap, G, D, device=device, generator=gen).half())
                  for _ in range(L)]

        def run():
            for k, v in caches:
                F_.decode_attention_rope(qkv, table, pos, widx, k, v, mask, H, G, D, split=cap > 256)
        out[f"capacity_{cap}_us"] = round(_graph_time(torch, device, run, reps=10) / L * 1e3, 2)
        if cap == 256:
            layers = [_w4_layer(torch, device, 4096, 4096, False, gen) for _ in range(L)]

            def pair():
                for (k, v), l in zip(caches, layers):
                    nxt = (l.prepare()._packed, _lib.NEXT_W4G32_PACKED, 4096, 4096)
                    a = F_.decode_attention_rope(qkv, table, pos, widx, k, v, mask, H, G, D, prefetch=nxt)
                    with torch.no_grad():
                        l(a)
            out["capacity_256_with_o_proj_us"] = round(_graph_time(torch, device, pair, reps=10) / L * 1e3, 2)
            del layers
        del caches
    out["note"] = ("launch to launch; round 2's kernel: 6.30 / 9.2 - 9.5 us; attention + o_proj 10.3 (profiles/r05_attention_ab.txt); what the launch "
                   "is made of: profiles/r05_attention_timeline.txt")
    return out


def prefill_attention(torch, device):
    """The attention of config 5's prefill (batch 4, one chunk of 2048 positions, 32 heads x 128, 2 key / value groups): the
    one-launch kernel (qlinear_prefill_attention) beside the GEMM route (two batched GEMMs around qlinear_masked_softmax); FLOPs
    count only visible (query, key) pairs."""
    from chatglm_q_amd import fused_ops as F_
    from chatglm_q_amd import model as M
    H, G, D, B, S = 32, 2, 128, 4, 2048
    dt = torch.float16
    gen = torch.Generator(device=device).manual_seed(5)
    q = torch.randn(B, S, H * D, device=device, generator=gen).to(dt)
    k = torch.randn(B, S, G, D, device=device, generator=gen).to(dt)
    v = torch.randn(B, S, G, D, device=device, generator=gen).to(dt)
    t = torch.arange(S, device=device)
    mask = ((t[None, None, :] > t[None, :, None]).expand(B, S, S).float() * -1e10).contiguous()
    flags = F_.attention_tile_flags(mask)
    attn = M.ChatGLM2Attention(H * D, H, D, G, 0, dtype=dt)
    flops = 4.0 * B * H * D * (S * (S + 1) // 2)

    def timed(fn, n):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    one = timed(lambda: F_.prefill_attention(q, k, v, mask, flags, S, H, G, D), 20)
    two = timed(lambda: attn.core(dt, q.view(B, S, G, H // G, D), k, v, mask), 5)
    a, b = F_.prefill_attention(q, k, v, mask, flags, S, H, G, D), attn.core(dt, q.view(B, S, G, H // G, D), k, v, mask)
    return {"workload": "causal attention of one prefill layer, batch 4 x 2048 positions, fp16",
            "one_launch_ms": round(one, 4), "one_launch_TFLOPs_visible_pairs": round(flops / one / 1e9, 1),
            "frac_of_2.5PF_f16_mfma": round(flops / one / 1e9 / 2500.0, 4),
            "gemm_route_ms": round(two, 4), "gemm_route_TFLOPs_visible_pairs": round(flops / two / 1e9, 1),
            "rel_l2_between_routes": float((a.float() - b.float()).norm() / b.float().norm())}


def int8_model_prefill(torch, device):
    """The int8 model family at BASELINE config 5's workload (4 x 2048 positions, one pass): weight-only (the reference's int8 forward)
    beside int8 activations (module.act_quant: chatglm_q/int8/qlinear.py:56-62 - the north_star's 'MFMA where activations are
    pre-quantized to int8': quantising producers + the int8 x int8 ring GEMM)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import int8_prefill
    r = int8_prefill.run(torch, device)
    flops = 2.0 * 4 * 2048 * (4096 * 4608 + 4096 * 4096 + 4096 * 27392 + 13696 * 4096) * 28
    for k in ("weight_only", "int8_activations"):
        r[k]["linear_Tops_per_s_if_all_time_were_linear"] = round(flops / r[k]["seconds"] / 1e12, 1)
    r["workload"] = "ChatGLM2-6B int8 per-channel weights, prefill batch 4 x 2048 (8192 rows per QLinear call), fp16, synthetic weights"
    return r


def run(torch, device):
    out = {}
    t0 = time.perf_counter()
    for name, fn in [("token_sweep", token_sweep), ("decode_shapes", per_shape), ("w8_decode", w8_decode),
                     ("w8a8_config3", w8a8_config3),
                     ("prefill_gemm_M8192", prefill_gemm), ("prefill_gemm_power", prefill_gemm_power), ("fp32_rows_512", fp32_many_rows), ("int8_prefill_gemm_M8192", int8_prefill_gemm),
                     ("prefill_attention_b4_s2048", prefill_attention), ("decode_attention", decode_attention), ("e2e_generate", e2e_generate), ("decode_shapes_bf16", per_shape_bf16), ("e2e_generate_bf16", e2e_generate_bf16), ("e2e_generate_int8", e2e_generate_int8),
                     ("int8_model_prefill", int8_model_prefill),
                     ("e2e_cpu", e2e_cpu)]:
        try:
            out[name] = fn(torch, device)
        except Exception as e:      # keep going: extras are informative only
            out[name] = {"error": repr(e)}
        torch.cuda.empty_cache()
    try:        # BASELINE config 5 as a recomputable fraction: the QLinear flops of the 4 x 2048 prefill over its wall time
        cp = out["e2e_generate"]["chunked_prefill_seq2048_batch4"]
        flops = 2.0 * 4 * 2048 * (4096 * 4608 + 4096 * 4096 + 4096 * 27392 + 13696 * 4096) * 28
        out["prefill_roofline"] = {"workload": "ChatGLM2-6B int4g32 prefill, batch 4 x 2048 positions, one pass of 8192 rows per QLinear call",
                                   "flops": flops, "seconds": cp["seconds"], "TFLOPs": round(flops / cp["seconds"] / 1e12, 1),
                                   "frac_of_2.5PF": round(flops / cp["seconds"] / 1e12 / 2500.0, 4),
                                   "note": "linear-layer flops only (attention, norms, rotary and lm_head run inside the same wall time)"}
    except (KeyError, TypeError):
        pass
    out["extras_seconds"] = round(time.perf_counter() - t0, 1)
    return out
