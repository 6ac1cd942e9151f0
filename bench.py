#!/usr/bin/env python3
"""Headline benchmark: slides/s of the DGDM hot path (pretrain_step forward + backward) on
synthetic tissue graphs, MI355X HIP path.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1] per GPU; configs[2] is the same per-GPU work at 8 GPUs =>
weak scaling): DGDM-Base (node_features=768, hidden=[512,256,128], T=10, heads=8, all defaults of
DGDMModel incl. dropout 0.1, spatial attention, graph U-Net, attention pooling), training mode,
batch of 4 synthetic graphs of 10 000 nodes / 50 000 directed edges per GPU, edge_attr [E,32],
pos [N,2]; inputs resident in HBM before the timed region.  One step = entity masking +
forward(pretrain) + backward of diffusion_loss (+ RCCL all-reduce of the flat gradient buffer when
N > 1) + AdamW step.  fp32 throughout.

Prints ONE JSON line (rank 0).  Extra objects: `roofline` (dominant kernel, HIP events on the
launch stream inside the timed region), `gather_roofline` (the north star's message-passing
gather at 10k x 768), `cpu_baseline` (the CPU oracle = port of the reference path, timed on the
host cores on a bounded sample; rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NODES, EDGES, FEATS, PER_GPU_BATCH = 10000, 50000, 768, 4
MODEL_CFG = dict(node_features=FEATS, hidden_dims=[512, 256, 128], num_diffusion_steps=10, attention_heads=8)
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md, dense fp32 matrix peak
FP16_MFMA_PEAK_TFLOPS = 2516.6  # MI355X_MICROARCH.md, dense fp16 matrix peak
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md, HBM3E spec
# what the PMC passes count (TCC_EA0_RDREQ / WRREQ and FETCH/WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes): bytes that left
# the L2 towards the fabric.  Reads served by the 256 MiB Infinity Cache are INCLUDED -- an upper bound of what HBM itself moved.
TRAFFIC_KIND = "fabric bytes per launch (L2 misses: Infinity-Cache hits included, so an upper bound of the HBM bytes)"


def _latest_profile(stem):
    """profiles/rNN_<stem>.json of the highest round that has one (the PMC passes are taken with rocprofv3 on the same bench command
    and committed; bench.py only reads them)."""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{stem}.json")))
    return c[-1] if c else None


PMC_TRAFFIC = _latest_profile("pmc_traffic")   # tools/pmc_traffic.py, separate --pmc passes
PMC_VALU = _latest_profile("pmc_valu")         # tools/pmc_valu.py, one --pmc pass
PMC_TRAFFIC_FP32 = _latest_profile("fp32_pmc_traffic")   # the same passes of `bench.py --precision fp32` (the strict_fp32 leg's kernels)
PMC_VALU_FP32 = _latest_profile("fp32_pmc_valu")
PMC_TRAFFIC_LARGE = _latest_profile("large_pmc_traffic")   # the same passes of `bench.py --large` (configs[3])
PMC_VALU_LARGE = _latest_profile("large_pmc_valu")
PMC_GATHER = _latest_profile("gather_pmc_traffic")       # tools/profile_gather.sh: the north-star gather microbenchmark, warm and cold


def arithmetic_error_note():
    """The sentence of the `dtype` field that says how far the default arithmetic sits from float64, built from the committed
    report it cites (tools/arithmetic_error_report.py -> profiles/rNN_arithmetic_error_vs_float64.txt: the `ALL` row of every
    case); empty when no report is committed -- the line never carries a literal that no file backs."""
    import glob
    import re
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_arithmetic_error_vs_float64.txt")))
    if not c:
        return ""
    case, rows = None, []
    for ln in open(c[-1]):
        m = re.match(r"# case (\w+):", ln)
        if m:
            case = m.group(1)
        m = re.match(r"ALL\s+\d+\s+(\S+) \| \S+\s+(\S+) \| \S+\s+(\S+) \|", ln)
        if m and case:
            rows.append(f"{case} {m.group(1)} / {m.group(2)} / {m.group(3)}")
    if not rows:
        return ""
    return ("; worst rel-L2 of any live parameter gradient of one step against the float64 oracle, default arithmetic / HIP kernels with "
            "fp32 operands / torch fp32 on the CPU: " + ", ".join(rows) + f" ({os.path.relpath(c[-1], ROOT)}, held by tests/test_hip_model.py::"
            "test_default_arithmetic_is_at_the_error_level_of_fp32)")


def attention_flops(num_graph_nodes, heads, head_dim, products):
    """2*N^2*H*d FLOP per QK^T-sized product, `products` of them per kernel."""
    return sum(2.0 * n * n * heads * head_dim * products for n in num_graph_nodes)


def _pmc_kernel(path, kernel):
    """Entry of `kernel` in a committed PMC summary; template arguments may be spelled with or without blanks."""
    try:
        with open(path) as f:
            ks = json.load(f)["kernels"]
    except (OSError, KeyError, ValueError, TypeError):
        return None
    want = kernel.replace(" ", "")
    for name, v in ks.items():
        if name.replace(" ", "") == want:
            return v
    return None          # a kernel that was not profiled has NO counter figure (never another instantiation's)


def _fabric_bytes(entry):
    """Bytes per launch of a PMC summary entry (`fabric_bytes_per_launch`; summaries of rounds 1-4 call the same figure
    `hbm_bytes_per_launch`)."""
    return entry.get("fabric_bytes_per_launch", entry.get("hbm_bytes_per_launch"))


def pmc_traffic(kernel, path=None):
    """HBM bytes per launch of `kernel` from the committed PMC passes (same command, same sizes), or None."""
    v = _pmc_kernel(path or PMC_TRAFFIC, kernel)
    return None if v is None else _fabric_bytes(v)


def pmc_valu(kernel, path=None):
    """VALU / matrix-pipe occupancy and the wave-cycle split of `kernel` from the committed PMC pass, or None."""
    path = path or PMC_VALU
    v = _pmc_kernel(path, kernel)
    if v is None:
        return None
    keep = ("valu_busy", "mfma_busy", "wave_cycles_active", "wave_cycles_issue_stalled", "wave_cycles_parked", "valu_share_of_active",
            "valu_insts_per_launch", "mfma_insts_per_launch", "occupancy_waves_per_simd", "kernel_cycles_per_launch")
    return dict({k: v[k] for k in keep if k in v}, source=f"committed PMC pass {os.path.relpath(path, ROOT)}, kernel {kernel}")


def gather_bytes(n, e, c):
    ent = e + n
    return ent * c * 4 + n * c * 4 + ent * 8 + (n + 1) * 4


def gather_unique_bytes(n, e, c):
    """Cold-cache lower bound of the same launch: every table row read once, every output row written once, the index arrays."""
    return n * c * 4 + n * c * 4 + (e + n) * 8 + (n + 1) * 4


def gather_microbench(dev, iters=200, cold_iters=12):
    """The north star's message-passing gather (one graph convolution's aggregation at 10k nodes x 768 features).
    `achieved` / `frac`: SURVEY.md 8(d) algorithmic bytes over the mean of back-to-back launches -- the 30.7 MB table then lives
    in L2 / Infinity Cache, so this is a cache-resident figure (it can exceed what HBM alone delivers).  `cold`: single launches,
    each after a 1 GiB buffer was rewritten (L2 and the 256 MiB Infinity Cache evicted), priced both on the algorithmic bytes and
    on the unique bytes (what HBM has to move at least)."""
    from dgdm_histopath_lab_amd import GraphStructure, ops
    from dgdm_histopath_lab_amd.synthetic import synthetic_graph
    g = synthetic_graph(0, NODES, EDGES, 8)
    gs = GraphStructure(g.edge_index.to(dev), NODES)
    x = torch.randn(NODES, FEATS, device=dev)
    y = torch.empty_like(x)
    for _ in range(20):
        ops.spmm_raw(gs.rowptr, gs.col, gs.w, x, NODES, out=y)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters):
        ops.spmm_raw(gs.rowptr, gs.col, gs.w, x, NODES, out=y)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / iters
    by, ub = gather_bytes(NODES, EDGES, FEATS), gather_unique_bytes(NODES, EDGES, FEATS)
    evict = torch.zeros(256 * 1024 * 1024, dtype=torch.float32, device=dev)       # 1 GiB
    cold = []
    for _ in range(cold_iters):
        evict.add_(1.0)
        a.record()
        ops.spmm_raw(gs.rowptr, gs.col, gs.w, x, NODES, out=y)
        b.record(); torch.cuda.synchronize()
        cold.append(a.elapsed_time(b) * 1e3)
    del evict
    cold.sort()
    cus = cold[len(cold) // 2]
    kname = "k_spmm<64, 3, 4, false>"
    warm_pmc, cold_pmc = _pmc_kernel(PMC_GATHER, kname), None
    try:
        with open(PMC_GATHER) as f:
            cold_pmc = json.load(f)["cases"]["cold"].get(kname)
    except (OSError, KeyError, ValueError, TypeError):
        pass
    src = None if PMC_GATHER is None else f"committed PMC passes {os.path.relpath(PMC_GATHER, ROOT)} (tools/profile_gather.sh), kernel {kname}"
    traffic = None if warm_pmc is None else _fabric_bytes(warm_pmc)
    return {"kernel": "dgdm_spmm (k_spmm<64,3,4,false>)", "workload": f"{NODES} nodes x {FEATS} feat, {EDGES}+{NODES} entries",
            "bound": "hbm", "achieved": round(by / us / 1e3, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(by / us / 1e3 / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": src,
            "traffic_over_algorithmic": None if traffic is None else round(traffic / by, 3),
            "traffic_over_unique": None if traffic is None else round(traffic / ub, 3),
            "cold_traffic": None if cold_pmc is None else _fabric_bytes(cold_pmc), "traffic_kind": TRAFFIC_KIND,
            "us_per_launch": round(us, 2), "algorithmic_bytes": by, "unique_bytes": ub,
            "note": "back-to-back launches: the table is re-read from L2 / Infinity Cache (cache-resident figure)",
            "unique_bytes_GBps": round(ub / us / 1e3, 1), "unique_frac": round(ub / us / 1e3 / HBM_PEAK_GBPS, 4),
            "cold": {"us_per_launch": round(cus, 2), "launches": cold_iters, "evicted_with": "1 GiB buffer rewritten before every launch",
                     "algorithmic_GBps": round(by / cus / 1e3, 1), "algorithmic_frac": round(by / cus / 1e3 / HBM_PEAK_GBPS, 4),
                     "unique_GBps": round(ub / cus / 1e3, 1), "unique_frac": round(ub / cus / 1e3 / HBM_PEAK_GBPS, 4)}}


def projection_microbench(dev, iters=50):
    """The largest dense contraction of the step (FeatureEncoder first layer: [4 x 10k, 768] x [512, 768]^T) on the shipped GEMM
    kernel, priced on the matrix pipe it runs on: algorithmic FLOP (2 M N K), the FLOP the kernel issues (3 fp16 MFMAs per product
    term for the fp16 hi+lo kernel, 6 bf16 MFMAs for the exact bf16 split), and the fp32-equivalent fraction."""
    from dgdm_histopath_lab_amd import ops
    m, k, n = NODES * PER_GPU_BATCH, FEATS, 512
    x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) / k ** 0.5; b = torch.randn(n, device=dev)
    math = ops.GEMM_MATH
    y = torch.empty(m, n, device=dev)
    for _ in range(5):
        ops.gemm_nt_raw(x, w, b, out=y, math=math)
    # as in the step: the launches are recorded into a HIP graph and replayed (through the Python wrapper a launch costs ~15 us of
    # host time, more than the device needs between two of these kernels)
    ops.WEIGHT_IMAGES.prepare(torch.device(dev))
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(iters):
            ops.gemm_nt_raw(x, w, b, out=y, math=math)
    g.replay()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    g.replay()
    e.record(); torch.cuda.synchronize()
    us = a.elapsed_time(e) * 1e3 / iters
    fl = 2.0 * m * k * n
    tf = fl / us / 1e6
    info = {"f16x2": ("dgdm_gemm_rows_img (k_gemm_img8<false>: weight pre-split into an fp16 hi+lo image, activation rows straight into MFMA fragments)"
                      if ops.USE_WEIGHT_IMAGES else "dgdm_gemm_nt_f16x2 (k_gemmh_rows<true,false,false>)",
                      "f16 dense matrix pipe (v_mfma_f32_32x32x16_f16)",
                      "fp16 hi+lo operands (power-of-two scaled by the operand maximum), 3 MFMAs per product, fp32 accumulate", 3),
            "bf16x3": ("dgdm_gemm_nt_bf16x3 (k_gemm3_rows<true,false,false>)", "bf16 dense matrix pipe (v_mfma_f32_32x32x16_bf16)",
                       "bf16 x3 exact split, 6 MFMAs per product, fp32 accumulate", 6),
            "fp32": ("dgdm_gemm_nt (k_gemm_rows)", "fp32 matrix pipe", "fp32", 1)}[math]
    peak = FP32_MFMA_PEAK_TFLOPS if math == "fp32" else FP16_MFMA_PEAK_TFLOPS      # bf16 and f16 dense peaks are equal
    out = {"kernel": info[0], "workload": f"[{m}, {k}] x [{n}, {k}]^T + bias", "bound": "mfma", "achieved": round(tf, 1), "peak": peak,
           "unit": "TFLOP/s", "frac": round(tf / peak, 4), "us_per_launch": round(us, 1), "algorithmic_flop": fl, "pipe": info[1],
           "timed_with": f"HIP events around one replay of a HIP graph of {iters} launches", "traffic": None}   # the committed PMC pass averages this kernel over all shapes of a step: not comparable per launch
    if math != "fp32":
        out.update({"mfma_dtype": info[2], "issued_tflops": round(info[3] * tf, 1), "issued_frac": round(info[3] * tf / peak, 4),
                    "fp32_equivalent_frac": round(tf / FP32_MFMA_PEAK_TFLOPS, 4)})
    return out


def host_cpu():
    """(model name, physical cores, logical cpus) of the host, from /proc/cpuinfo."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None:
                cores.add((phys, core)); phys = core = None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    return model, (len(cores) or logical), logical


def cpu_baseline(nodes, edges):
    """CPU oracle (restatement of the reference path incl. its dense attention and dropout, training mode), SURVEY.md 8(d)
    protocol: BASELINE configs[0] (ONE 2k-node / 8k-edge graph), 3 warm-up + 5 timed fwd+bwd steps, median -> `value`;
    plus ONE step on one graph of the headline size.  Threads = physical cores of this host."""
    from oracle import dgdm_oracle as O
    from dgdm_histopath_lab_amd.synthetic import synthetic_batch
    model, physical, logical = host_cpu()
    threads = max(1, min(physical, logical))
    torch.set_num_threads(threads)
    cfg = O.OracleConfig(**MODEL_CFG)
    P = O.init_params(cfg, seed=0)

    def step(n, e):
        b = synthetic_batch(0, 1, n, e, FEATS)
        idx = torch.randperm(n)[: int(0.15 * n)]
        t0 = time.perf_counter()
        O.loss_and_grads(P, cfg, b, mask_indices=idx, mask_token=torch.randn(FEATS), training=True)
        return time.perf_counter() - t0
    for _ in range(3):
        step(2000, 8000)
    ts = sorted(step(2000, 8000) for _ in range(5))
    med = ts[2]
    # BASELINE.md 2 "Reported: per-stage split": one more step of the same workload with the oracle's stage clock (forward wall
    # time per stage; the backward is one autograd pass and is reported whole)
    stages: dict = {}
    b = synthetic_batch(0, 1, 2000, 8000, FEATS)
    t0 = time.perf_counter()
    Pg = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    out = O.pretrain_step(Pg, cfg, b, mask_indices=torch.randperm(2000)[:300], mask_token=torch.randn(FEATS), training=True, stages=stages)
    t1 = time.perf_counter()
    out["total_pretrain_loss"].backward()
    t2 = time.perf_counter()
    fwd = sum(stages.values())
    split = {"forward_s": round(t1 - t0, 4), "backward_s": round(t2 - t1, 4),
             "forward_share": {k: round(v / fwd, 4) for k, v in stages.items()},
             "forward_seconds": {k: round(v, 4) for k, v in stages.items()},
             "stages": "graph_structure = CSR / degree normalisation; feature_encoder (a1); graph_encoder (a2-a4); spatial_attention "
                       "(a5-a7, dense [H,N,N] scores + dropout mask); graph_unet (a8-a9); diffusion (a10-a12); pool (a14)"}
    big = step(nodes, edges)
    return {"value": round(1.0 / med, 4), "unit": "slides/s", "cores": threads, "kind": "port", "stage_split": split,
            "sample": f"BASELINE configs[0]: 1 graph of 2000 nodes / 8000 edges, fwd+bwd (training mode, dropout 0.1), 3 warm-up + 5 timed "
                      f"steps, median {med:.3f} s (min {ts[0]:.3f}, max {ts[-1]:.3f}); CPU oracle = restatement of the reference path",
            "headline_size": {"value": round(1.0 / big, 5), "unit": "slides/s",
                              "sample": f"1 graph of {nodes} nodes / {edges} edges, 1 fwd+bwd step, {big:.1f} s"},
            "cpu": model, "physical_cores": physical, "logical_cpus": logical}


def sample_loop_bench(model, dev, rows=10000, reps=5):
    """DiffusionLayer.sample (reference: core/diffusion.py:214-275, the T-step denoise loop of the north star), graph-replayed, at
    `rows` x hidden_dims[-1] with 10 and 50 inference steps: ms per loop, the algorithmic bytes of the seven launches of a step against
    the HBM peak, and the CPU oracle's time for the same loop beside it (rank 0, N = 1; after the timed region of the headline)."""
    from oracle import dgdm_oracle as O        # cpu_baseline leg only
    dl = model.diffusion_layer
    C, Hd, T = dl.node_dim, dl.hidden_dim, dl.num_timesteps
    was_training = dl.training
    dl.eval()
    from dgdm_histopath_lab_amd import ops
    fused = ops.denoise_ddpm_step_supported(C) and Hd == 2 * C
    step_flop = 2.0 * rows * (C * 2 * Hd + 2 * Hd * Hd + Hd * C)
    if fused:
        # one launch per step (csrc/sample_step.hip): the activations never leave the CU.  HBM bytes of a step: the normal draw writes z,
        # the step reads x and z and writes x (fp32 rows) + the three weight images once (they stay in L2 afterwards)
        w_bytes = 4 * (C * 2 * Hd + 2 * Hd * Hd + Hd * C)
        step_bytes = rows * 4 * 4 * C + w_bytes
        launches, what = 1, ("ONE launch per step for the whole denoiser + DDPM update (csrc/sample_step.hip: 32 rows per workgroup through three "
                             "Linear layers, two GroupNorm + SiLU and the update, activations in LDS); the loop's normal draws in one launch up front")
    else:
        # per step: GEMM C->2Hd (read x, write h1), GroupNorm+SiLU (read + write h1), GEMM 2Hd->Hd, GroupNorm+SiLU, GEMM Hd->C, one normal
        # draw (write z), the DDPM update (read x, eps, z; write x): fp32 rows, weights (0.9 MB) not counted
        step_bytes = rows * 4 * ((C + 2 * Hd) + 2 * 2 * Hd + (2 * Hd + Hd) + 2 * Hd + (Hd + C) + C + 4 * C)
        launches, what = 7, "7 launches per step: 3 tile GEMMs, 2 fused GroupNorm+SiLU rows, one normal draw, one DDPM update"
    out = {"rows": rows, "width": C, "denoiser_widths": [C + Hd, 2 * Hd, Hd, C], "T": T, "launches_per_step": launches,
           "algorithmic_bytes_per_step": step_bytes, "flop_per_step": step_flop,
           "note": "sample(graphed=True): the whole loop is ONE recorded HIP graph (" + what + "; the time-embedding MLP runs once before the "
                   "loop); eval mode as generate()"}
    P = {k: v.detach().float().cpu() for k, v in model.state_dict().items() if k.startswith("diffusion_layer.") and "scheduler" not in k}
    sched = O.diffusion_schedule(T, MODEL_CFG.get("diffusion_schedule", "cosine"))
    try:
        for steps in (10, 50):
            for _ in range(2):
                dl.sample((rows, C), dev, num_inference_steps=steps, graphed=True)       # warm-up + recording
            torch.cuda.synchronize()
            g, sx = dl._sample_graphs[((rows, C), steps, str(dev), False, False)][:2]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            t0 = time.perf_counter()
            for _ in range(reps):
                dl.sample((rows, C), dev, num_inference_steps=steps, graphed=True)
            torch.cuda.synchronize()
            ms_call = (time.perf_counter() - t0) / reps * 1e3
            gbps = steps * step_bytes / (ms * 1e-3) / 1e9
            x_init = torch.randn(rows, C)
            noises = [torch.randn(rows, C) for _ in range(steps - 1)]
            torch.set_num_threads(min(32, os.cpu_count() or 8))
            with torch.no_grad():
                O.ddpm_sample(P, sched, T, x_init, noises, min(steps, 3))                    # warm-up
                t0 = time.perf_counter()
                O.ddpm_sample(P, sched, T, x_init, noises, steps)
                cpu_ms = (time.perf_counter() - t0) * 1e3
            out[f"steps{steps}"] = {"ms_per_loop": round(ms, 4), "us_per_step": round(ms / steps * 1e3, 2), "ms_per_call": round(ms_call, 4),
                                    "timed_with": f"HIP events around {reps} replays of the recorded loop; ms_per_call = wall clock of sample() "
                                                  "itself (input copy + replay + output clone)",
                                    "roofline": {"bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                                 "frac": round(gbps / HBM_PEAK_GBPS, 4), "traffic": None,
                                                 "mfma_tflops": round(steps * step_flop / (ms * 1e-3) / 1e12, 1),
                                                 "mfma_frac": round(steps * step_flop / (ms * 1e-3) / 1e12 / FP16_MFMA_PEAK_TFLOPS, 4),
                                                 "note": ("one launch of %d workgroups per step on 256 CUs: a %d-row problem is 1.2 rounds of workgroups, "
                                                          "each a chain of three dependent layers -- latency, neither bytes nor FLOP" % ((rows + 31) // 32, rows))
                                                         if fused else "7 dependent launches of 16-41 MB each per step: start-up and drain of each "
                                                         "launch, not bytes, are what a step costs at this size"},
                                    "cpu_baseline": {"value": round(cpu_ms, 2), "unit": "ms per loop", "cores": torch.get_num_threads(), "kind": "port",
                                                     "sample": f"the same loop ({steps} steps, {rows} rows) by oracle.ddpm_sample, once"},
                                    "speedup_vs_cpu": round(cpu_ms / ms, 1)}
    except Exception as e:          # never lose the headline line to this leg
        out["error"] = f"{type(e).__name__}: {e}"[:300]
    finally:
        dl.train(was_training)
    return out


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks as FRESH child processes through
    `python -m torch.distributed.run` (one process per GPU, rendezvous on 127.0.0.1 at a port found free by bind(0)), hand rank 0's
    ONE JSON line through to stdout and return the launcher's exit code.  Must run before this process touches the GPU: no
    `torch.cuda.is_available()`, no import of the kernels -- `torch.cuda.device_count()` does not initialise HIP on this image --
    and the children are started with subprocess (fork + exec of a process that has NOT initialised HIP), never by replacing
    this process."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and os.environ.get("DGDM_BENCH_ONE_DEVICE") != "1":
        print(f"bench.py: --gpus {n} but this node shows {have} GPU(s)", file=sys.stderr)
        return 2
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    # (own session: if the ranks hang -- a collective that never completes -- the whole group can be ended; stderr is inherited:
    # progress and errors stay visible)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, start_new_session=True)
    limit = float(os.environ.get("DGDM_BENCH_RANK_TIMEOUT", "1500"))
    import threading
    timed_out = []

    def _expire():
        timed_out.append(True)
        print(f"bench.py: the {n} ranks did not finish within {limit:.0f} s (DGDM_BENCH_RANK_TIMEOUT); ending them", file=sys.stderr, flush=True)
        try:
            os.killpg(proc.pid, 15)
            time.sleep(10)
            os.killpg(proc.pid, 9)
        except ProcessLookupError:
            pass
    timer = threading.Timer(limit, _expire)
    timer.daemon = True
    timer.start()
    lines = []
    for ln in proc.stdout:
        if ln.strip():
            lines.append(ln.rstrip("\n"))
    rc = proc.wait()
    timer.cancel()
    if timed_out:
        rc = rc or 124
    result = [ln for ln in lines if ln.lstrip().startswith("{")]
    for ln in lines:                 # anything a rank printed beside the result goes to stderr: stdout carries ONE line
        if not result or ln is not result[-1]:
            print(ln, file=sys.stderr)
    if result:
        print(result[-1], flush=True)
    elif rc == 0:
        print("bench.py: the ranks exited without printing a result line", file=sys.stderr)
        rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=PER_GPU_BATCH, help="slides per GPU")
    ap.add_argument("--nodes", type=int, default=NODES)
    ap.add_argument("--edges", type=int, default=EDGES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-strict", action="store_true", help="skip the strict-fp32-attention leg")
    ap.add_argument("--no-raster", action="store_true", help="skip the raster-pixel-positions leg (reported beside the headline)")
    ap.add_argument("--no-sample-loop", action="store_true", help="skip the sample_loop object (the T-step denoise loop, graph-replayed)")
    ap.add_argument("--eval-mode", action="store_true", help="dropout off (diagnostics only; not the headline)")
    ap.add_argument("--precision", choices=["default", "fp32"], default="default",
                    help="fp32: run the MAIN leg with fp32 operands on the fp32 matrix instructions in the attention and in every dense "
                         "layer (what the `strict_fp32` object of the default run reports; used to take its rocprof / PMC profiles)")
    ap.add_argument("--mixed", action="store_true",
                    help="BASELINE configs[4] shape instead of the headline: a stream of batches whose graphs have 1k..10k nodes "
                         "(E = 5 N), 8 different batches resident in HBM and cycled; reported under config.workload, not comparable "
                         "with the fixed-size number")
    ap.add_argument("--large", action="store_true",
                    help="BASELINE configs[3] instead of the headline: the Large model (hidden 1024/512/256, 16 heads, T=20) on 50k-node / "
                         "300k-edge graphs, 1 graph per GPU unless --batch is given; reported under config.workload")
    ap.add_argument("--pixel-positions", type=float, default=0.0, metavar="PITCH",
                    help="NOT the headline: replace the synthetic positions (U[0,1)^2, BASELINE's) by patch centres of a slide scanned row "
                         "by row at PITCH pixels (what the reference's preprocessing stores); with temperature 1 all but a band of the "
                         "attention's block pairs are exact zeros and the zero-block map walks them over; reported under config.workload")
    ap.add_argument("--sustain-seconds", type=float, default=6.0,
                    help="N=1 only: after the timed K steps, keep replaying the same step for this many seconds and report the rate as "
                         "`sustained` (steady-state clocks; also gives an external GPU-activity sampler a window of pure GPU work well "
                         "before the CPU baseline starts); 0 switches it off")
    ap.add_argument("--eager", action="store_true",
                    help="launch every kernel of the step from the host instead of replaying the step from HIP graphs "
                         "(training.GraphedPretrainStep, the default for the fixed-shape headline workload)")
    args = ap.parse_args()
    if args.eager or args.mixed:
        # eager launches: kernel arguments in device memory (+4-5 % when the step is GPU-bound; it costs host time per launch, so
        # it is not forced on steps of tiny graphs, and a recorded step does not care).  Read by the HIP runtime at its first call.
        os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher (the reference gets its ranks from
        # pl.Trainer(accelerator="gpu", devices=gpus), cli/train.py:346-359).  Nothing here has touched the GPU yet.
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} in the environment (a launcher's world size must equal "
                  f"--gpus; without WORLD_SIZE this script starts its own ranks)", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (the HIP path has no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    # rehearsal knobs (not used by the driver): all ranks on one GPU over gloo, to exercise the N > 1 code path on a 1-GPU box
    one_device = os.environ.get("DGDM_BENCH_ONE_DEVICE") == "1"
    backend = os.environ.get("DGDM_BENCH_DIST_BACKEND", "nccl")
    force_dist = os.environ.get("DGDM_BENCH_FORCE_DIST") == "1"     # one rank, but through RCCL and the split recording
    dev_index = 0 if one_device else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    import torch.distributed as dist
    from dgdm_histopath_lab_amd import DGDMModel, ops
    if args.precision == "fp32":
        ops.configure(attention="fp32", gemm="fp32")
    from dgdm_histopath_lab_amd.parallel import FlatGradAllReducer
    from dgdm_histopath_lab_amd.synthetic import synthetic_batch

    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL prints a version banner on STDOUT when its first communicator comes up; the contract of this script is ONE JSON
        # line on stdout, so file descriptor 1 points at stderr until the communicator exists (first collective included)
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(backend)
            warm = torch.zeros(1, device=dev)
            dist.all_reduce(warm)
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    cfg = dict(MODEL_CFG)
    if args.large:
        cfg.update(hidden_dims=[1024, 512, 256], attention_heads=16, num_diffusion_steps=20)
        if args.nodes == NODES and args.edges == EDGES:
            args.nodes, args.edges = 50000, 300000
        if args.batch == PER_GPU_BATCH:
            args.batch = 1
    torch.manual_seed(0)
    model = DGDMModel(**cfg).to(dev)
    model.train(not args.eval_mode)
    # training/trainer.py:221-226 defaults (AdamW, lr 1e-4, weight decay 1e-5); optim.DGDMAdamW: torch's AdamW arithmetic in one launch
    from dgdm_histopath_lab_amd.optim import DGDMAdamW
    opt = DGDMAdamW(model.parameters(), lr=1e-4, weight_decay=1e-5)
    reducer = FlatGradAllReducer(model, world, always=force_dist) if (world > 1 or force_dist) else None
    # rank r owns slides [r*B, (r+1)*B): independent units, no data-path collective
    batch = synthetic_batch(rank * args.batch, args.batch, args.nodes, args.edges, FEATS).to(dev)
    sizes = [args.nodes] * args.batch
    if args.pixel_positions > 0:
        import math as _m
        w = int(_m.ceil(_m.sqrt(args.nodes)))
        i = torch.arange(args.nodes, device=dev)
        one = torch.stack([(i % w).float(), (i // w).float()], 1) * args.pixel_positions
        batch.pos = one.repeat(args.batch, 1).contiguous()
        batch.pos_extent = float(w * args.pixel_positions)
    stream, balance_note = None, None
    if args.mixed:
        # BASELINE configs[4]: per step a global pool of world x batch graphs with N ~ U{1k..10k}, E = 5 N, assigned to ranks by
        # the cost-aware LPT sharding (parallel.balance_slides over parallel.slide_cost): attention cost grows with N^2, so equal
        # COUNTS per rank would leave the ranks unbalanced.  Every rank draws the same pool (same seed) and keeps its own bin.
        from dgdm_histopath_lab_amd.parallel import BalancedSlideLoader
        from dgdm_histopath_lab_amd.synthetic import synthetic_graph
        g = torch.Generator().manual_seed(77)
        ns = torch.randint(1000, 10001, (8 * world * args.batch,), generator=g).tolist()
        pool = [synthetic_graph(i, n, 5 * n, FEATS) for i, n in enumerate(ns)]
        loader = BalancedSlideLoader(pool, world * args.batch, world, rank, device=dev)
        stream = list(loader)                      # 8 per-rank batches, resident in HBM, cycled
        balance_note = {"sharding": "parallel.BalancedSlideLoader (LPT on alpha N^2 + beta N + gamma E per step)",
                        "max_over_mean_rank_load": round(loader.max_over_mean_load(), 4)}
        step_no = [0]

    def step():
        opt.zero_grad(set_to_none=True)
        cur = batch
        if stream is not None:
            cur = stream[step_no[0] % len(stream)]
            step_no[0] += 1
        out = model.pretrain_step(cur, mask_ratio=0.15)
        with ops.deferred_weight_grads():      # zero_grad(set_to_none=True) above: the dW reductions of the pass run in one launch
            out["total_pretrain_loss"].backward()
        if reducer is not None:
            reducer.all_reduce()
        opt.step()
        return out["total_pretrain_loss"]

    graph_note = None
    eager_step = step
    if not args.eager and stream is not None:
        # the 8 layouts of the stream recur: one recording per layout (training.GraphedStepCache); with a reducer every recording is
        # split around ONE all-reduce of the same flat gradient buffer (warm-up steps of a layout exchange the same single message)
        from dgdm_histopath_lab_amd.training import GraphedStepCache
        cache = GraphedStepCache(model, opt, mask_ratio=0.15, max_layouts=len(stream), grad_reducer=reducer)

        def step():
            cur = stream[step_no[0] % len(stream)]
            step_no[0] += 1
            return cache(cur)
        for _ in range(len(stream) * (cache.warmup + 2)):     # every layout primed and recorded: setup, not part of the W warmup steps
            step()
        graph_note = (f"HIP graph replay, one recording per layout ({len(cache.steps)} layouts, training.GraphedStepCache"
                      + (", each split around the gradient all-reduce" if reducer is not None else "") + ")")
    elif not args.eager and stream is None:
        from dgdm_histopath_lab_amd.training import GraphedPretrainStep
        gstep = GraphedPretrainStep(model, opt, mask_ratio=0.15, grad_reducer=reducer)
        # the batch is resident in HBM before the timed region (contract of this script): after the first call it lives in the
        # recording's own input buffers (GraphedPretrainStep.input_buffers, where a loader's host-to-device copy would put it) and
        # is handed over from there -- no device-to-device copy of 130 MB of inputs per step that the reference's step does not have
        graph_step = (lambda: gstep(gstep.input_buffers if gstep.input_buffers is not None else batch))
        done = 0
        try:
            for _ in range(gstep.warmup + 1):     # eager priming + recording: setup, not part of the W warmup steps
                graph_step()
                done += 1
        except Exception as e:                    # recording refused on this box: finish the priming steps eagerly (every rank
            graph_note = f"eager (recording failed: {type(e).__name__}: {e})"[:300]   # issues the same collectives) and stay eager
            print(f"bench.py: rank {rank}: {graph_note}", file=sys.stderr)
            for _ in range(gstep.warmup + 1 - done):
                eager_step()
        ok = torch.tensor([0.0 if graph_note else 1.0], device=dev)
        if world > 1 or force_dist:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) > 0:
            step = graph_step
        else:
            graph_note = graph_note or "eager (another rank could not record the step)"
            args.eager = True

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timed = ["attn_fwd", "attn_bwd_dq", "attn_bwd_dkv", "attn_bwd_fused", "attn_bwd_dq_reduce", "spmm_c512"]
    TSTAT = "mean" if stream is not None else "median"      # fixed shapes: the median launch (robust against one host hiccup); mixed stream: total / count
    graphed = step is not eager_step
    live_sink = None
    if not graphed:
        ops.TIMERS.start(timed)
        ops.ATTN_SKIP_MAP_SINK = live_sink = []      # a list append per attention call; the maps are counted after the timed region
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    t_local = None
    if world > 1:
        torch.cuda.synchronize()
        t_local = time.perf_counter() - t0      # this rank's own work done (its GPU idle), before it waits for the others
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ops.ATTN_SKIP_MAP_SINK = None
    eager_timed_steps = args.steps
    if graphed and (stream is None or world == 1):     # (the mixed stream's extra eager pass: one rank only -- N > 1 reports no roofline)
        # kernels launched by a graph replay cannot be bracketed by events: the per-kernel durations of the roofline come
        # from eager launches of the same step right after the timed region (same kernels, same shapes, same clocks); the mixed
        # stream runs each of its layouts once.  The zero-block maps of those launches are kept: the roofline counts the scores
        # the kernels EVALUATED (all of them on BASELINE's positions), not N^2 regardless
        eager_timed_steps = len(stream) if stream is not None else min(args.steps, 5)
        if stream is not None:
            step_no[0] = 0
        ops.ATTN_SKIP_MAP_SINK = live_sink = []
        ops.TIMERS.start(timed)
        for _ in range(eager_timed_steps):
            eager_step()
        torch.cuda.synchronize()
        ops.ATTN_SKIP_MAP_SINK = None
    ops.TIMERS.stop()

    def live_scores(sink):
        """(forward, one-pass backward, all) scores summed over the attention calls of `sink` (one host read per call: after the timed region)."""
        f = b = t = 0
        for m, plan_, H_ in sink:
            a = ops.attn_skip_live_scores(m, plan_, H_)
            f, b, t = f + a[0], b + a[1], t + a[2]
        return f, b, t
    pairs = None
    if live_sink:
        f, b, t = live_scores(live_sink)
        pairs = {"attn_fwd": f / max(t, 1), "attn_bwd_fused": b / max(t, 1), "scores_all": t, "scores_forward": f, "scores_backward": b,
                 "attention_calls": len(live_sink)}
    sustained = None
    if world == 1 and args.sustain_seconds > 0:
        n_s = max(args.steps, int(args.sustain_seconds / max(dt / args.steps, 1e-4)) + 1)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(n_s):
            step()
        torch.cuda.synchronize()
        d_s = time.perf_counter() - t1
        sustained = {"value": round(args.batch * n_s / d_s, 3), "unit": "slides/s", "steps": n_s, "seconds": round(d_s, 2),
                     "ms_per_step": round(d_s / n_s * 1e3, 3), "note": "the same step, replayed back to back after the timed region"}
    # NOT the headline: the same recorded step on positions as the reference's preprocessing stores them (patch centres in pixels, a
    # row-by-row raster at 224-pixel pitch; preprocessing/tissue_graph_builder.py:381-384).  -distance / temperature then makes all but
    # a band of the attention weights 0.0f and the zero-block map (csrc/attn_skip.hip, computed inside the recording from whatever
    # positions the input buffers hold) walks those block pairs over.  The positions are written into the recording's input buffer,
    # the step replayed, the synthetic positions restored.
    raster = None
    if world == 1 and graphed and stream is None and not args.large and args.pixel_positions == 0 and not args.no_raster:
        import copy
        import math as _m
        from dgdm_histopath_lab_amd.training import GraphedPretrainStep
        w = int(_m.ceil(_m.sqrt(args.nodes)))
        i = torch.arange(args.nodes, device=dev)
        one = torch.stack([(i % w).float(), (i // w).float()], 1) * 224.0
        rbatch = copy.copy(batch)
        rbatch.pos = one.repeat(args.batch, 1).contiguous()
        rbatch.pos_extent = float(w * 224)      # known on the host, as a loader knows it: the recording holds the zero-block map's launches
        gr = GraphedPretrainStep(model, opt, mask_ratio=0.15)
        rin = lambda: gr.input_buffers if gr.input_buffers is not None else rbatch
        for _ in range(gr.warmup + 1 + 2):
            gr(rin())
        n_r = max(5, min(args.steps, 20))
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(n_r):
            gr(rin())
        torch.cuda.synchronize()
        d_r = (time.perf_counter() - t1) / n_r
        # the scores the kernels evaluated on these positions: one eager step on them with the maps kept (after the timing)
        keep_batch, batch = batch, rbatch
        ops.ATTN_SKIP_MAP_SINK = rs = []
        eager_step()
        torch.cuda.synchronize()
        ops.ATTN_SKIP_MAP_SINK = None
        batch = keep_batch
        rf, rb, rt = live_scores(rs)
        for _ in range(2):
            loss = step()
        torch.cuda.synchronize()
        raster = {"pairs_live": {"forward": round(rf / max(rt, 1), 4), "backward": round(rb / max(rt, 1), 4),
                                 "note": "fraction of the N^2 H scores the forward / the one-pass backward evaluated (unmarked block pairs of the "
                                         "zero-block map; the backward at key-super-block granularity)"},
                  "value": round(args.batch / d_r, 3), "unit": "slides/s", "ms_per_step": round(d_r * 1e3, 3), "steps": n_r,
                  "positions": "patch centres of a row-by-row raster at 224-pixel pitch (level-0 pixels, as the reference's preprocessing "
                               "stores them), temperature 1; pos_extent known on the host",
                  "note": "NOT the headline (BASELINE's positions are U[0,1)^2, where no block pair can be zero and the step runs without "
                          "the map): a second recording of the same step on these positions; block pairs of the attention whose weights are "
                          "exactly 0.0f are skipped with bit-identical results (ops.attn_zero_blocks_possible)"}
    # the same step at the reference's own arithmetic (fp32 operands on the fp32 matrix instructions, in the attention and in every
    # dense layer): measured here, in the same process on the same box, so the two numbers are comparable
    strict = None
    if world == 1 and stream is None and not args.large and not args.eager and not args.no_strict and ops.ATTN_PRECISION == "fp16x2":
        from dgdm_histopath_lab_amd.training import GraphedPretrainStep
        prev = ops.configure(attention="fp32", gemm="fp32")
        try:
            g32 = GraphedPretrainStep(model, opt, mask_ratio=0.15)
            in32 = lambda: g32.input_buffers if g32.input_buffers is not None else batch
            for _ in range(g32.warmup + 1 + 2):
                g32(in32())
            n32 = max(5, min(args.steps, 20))
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for _ in range(n32):
                g32(in32())
            torch.cuda.synchronize()
            d32 = (time.perf_counter() - t1) / n32
            strict = {"value": round(args.batch / d32, 3), "unit": "slides/s", "ms_per_step": round(d32 * 1e3, 3), "steps": n32,
                      "attention": "fp32 operands on v_mfma_f32_16x16x4_f32 (csrc/attn_fwd.hip, attn_bwd.hip)",
                      "dense_layers": "fp32 operands on v_mfma_f32_32x32x2_f32 (csrc/gemm.hip)"}
            # the dominant kernel of THIS leg, timed like the headline's: eager launches of the same step bracketed by HIP events
            main_timers = ops.TIMERS.summary(TSTAT)
            ops.TIMERS.start(timed)
            for _ in range(3):
                eager_step()
            torch.cuda.synchronize()
            ops.TIMERS.stop()
            strict["_timers"] = ops.TIMERS.summary(TSTAT)
            ops.TIMERS.events = {}
            strict["_main_timers"] = main_timers
        finally:
            ops.configure(**prev)
    per_rank = None
    if world > 1:
        per = [None] * world
        dist.all_gather_object(per, (float(dt), float(t_local)))      # any backend (the rehearsal runs over gloo)
        own = [v[1] for v in per]
        per_rank = {"ms_per_step_min": round(min(own) / args.steps * 1e3, 3), "ms_per_step_max": round(max(own) / args.steps * 1e3, 3),
                    "ms_per_step_by_rank": [round(v / args.steps * 1e3, 3) for v in own],
                    "note": "time until the rank's own GPU work of the K steps was done, before the closing barrier (every step holds one "
                            "gradient all-reduce, so ranks cannot drift apart by more than a step); `ms_per_step` is the barrier-bracketed maximum"}
        dt = max(v[0] for v in per)
    loss_val = float(loss.item())

    result = None
    if rank == 0:
        timers = strict.pop("_main_timers") if strict is not None else ops.TIMERS.summary(TSTAT)
        heads, hd = cfg["attention_heads"], 16
        # products of 2 N^2 H d FLOP each: forward S, PV; two-pass backward dQ: S, dP, dS K / dK,dV: S, dP, P^T dO, dS^T Q; one-pass
        # backward (default): S, dP, P^T dO, dS^T Q, dS K -- each score evaluated once
        if stream is not None:      # the mixed stream: FLOP of a step = mean over its layouts (the eager timing pass ran each once)
            def _sizes(b_):
                ptr_ = getattr(b_, "ptr", None)
                if ptr_ is not None:
                    pl = ptr_.tolist() if isinstance(ptr_, torch.Tensor) else list(ptr_)
                    return [pl[i + 1] - pl[i] for i in range(len(pl) - 1)]
                return torch.bincount(b_.batch).tolist()
            layout_sizes = [_sizes(b_) for b_ in stream]
            aflops = lambda k: sum(attention_flops(sz, heads, hd, k) for sz in layout_sizes) / len(layout_sizes)
        else:
            aflops = lambda k: attention_flops(sizes, heads, hd, k)
        flops = {"attn_fwd": aflops(2), "attn_bwd_dq": aflops(3), "attn_bwd_dkv": aflops(4), "attn_bwd_fused": aflops(5)}
        split = ops.ATTN_PRECISION == "fp16x2"
        k16 = {"attn_fwd": "k_attn_h_fwd<4,1,1,3>", "attn_bwd_dq": "k_attn_h_bwd_dq<4,1,1,1>", "attn_bwd_dkv": "k_attn_h_bwd_dkv<2,1,1,3>",
               "attn_bwd_fused": f"k_attn_h_bwd_fused<{0 if args.eval_mode else 1},2>"}
        dr = "false" if args.eval_mode else "true"
        k32 = {"attn_fwd": f"k_attn_fwd<4,64,{dr}>", "attn_bwd_dq": f"k_attn_bwd_dq<4,64,{dr}>", "attn_bwd_dkv": f"k_attn_bwd_dkv<4,32,{dr}>",
               "attn_bwd_fused": "(fp32 path has no one-pass backward)"}
        # MFMAs the split-fp16 kernels issue per algorithmic product: Q'K and dO V as [hi|lo].[hi|hi] + [hi|lo].[lo|lo] (2 instructions
        # of twice the reduction length: 4x the FLOP), P V / P^T dO / dS^T Q / dS K as hi.hi + lo.hi + hi.lo (3x), ones.P twice (forward)
        # ... the one-pass backward's dS K as two MFMAs whose 32 reduction slots hold 16 keys x {hi, lo} (4x the FLOP of a 16-deep product)
        issued_x = {"attn_fwd": (4 + 3 + 2 * 1.0) / 2, "attn_bwd_dq": (4 + 4 + 3) / 3, "attn_bwd_dkv": (4 + 4 + 3 + 3) / 4,
                    "attn_bwd_fused": (4 + 4 + 3 + 3 + 4) / 5}

        def attention_roofline(tm, names, fp16_pipe, graphed_note, valu_path=None, traffic_path=None, timed_steps=1, live=None, pmc=True):
            dom = max((k for k in flops if k in tm), key=lambda k: tm[k][1] * tm[k][0]) if any(k in tm for k in flops) else None
            if dom is None:
                return {"note": "no attention kernel was timed"}
            ms = tm[dom][1]
            # launches of the kernel per step (the one-pass backward runs once per scratch group: 1 at the headline batch)
            per_step = max(1.0, tm[dom][0] / max(1, timed_steps))
            # the FLOP of the scores the kernel EVALUATED: block pairs the zero-block map marks (exact zeros) are not work done
            live_frac = 1.0 if live is None else float(live.get(dom, 1.0))
            tf = flops[dom] * live_frac / per_step / (ms * 1e-3) / 1e12
            peak = FP16_MFMA_PEAK_TFLOPS if fp16_pipe else FP32_MFMA_PEAK_TFLOPS
            # ALGORITHMIC FLOP of the reference's products (2 N^2 H d each) per second of the dominant kernel,
            # priced against the dense peak of the matrix pipe the kernel RUNS ON
            mfma = {"bound": "mfma", "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4),
                    "pipe": "f16 dense matrix pipe (v_mfma_f32_16x16x32_f16)" if fp16_pipe else "fp32 matrix pipe (v_mfma_f32_16x16x4_f32)",
                    "algorithmic_flop": flops[dom] * live_frac / per_step, "launches_per_step": per_step,
                    "pairs_live": {"frac": round(live_frac, 4), "all_pairs_flop": flops[dom] / per_step,
                                   "note": "fraction of the N^2 H scores this kernel evaluated (unmarked block pairs of the zero-block map, counted on "
                                           "the device after the timed region); 1.0 on BASELINE's U[0,1)^2 positions"
                                           if live is not None else "no zero-block map on this path: every pair evaluated"}}
            if fp16_pipe:
                mfma.update({"mfma_dtype": "every operand (Q', K, V, dO, P, dS) as fp16 hi+lo, fp32 accumulate",
                             "products": {"attn_fwd": "S, PV", "attn_bwd_dq": "S, dP, dS K", "attn_bwd_dkv": "S, dP, P^T dO, dS^T Q",
                                          "attn_bwd_fused": "S, dP, P^T dO, dS^T Q, dS K (one pass: dQ, dK, dV)"}[dom],
                             "issued_tflops": round(issued_x[dom] * tf, 1), "issued_frac": round(issued_x[dom] * tf / peak, 4),
                             "fp32_equivalent_frac": round(tf / FP32_MFMA_PEAK_TFLOPS, 4)})
            tpath = (traffic_path or (PMC_TRAFFIC_LARGE if args.large else PMC_TRAFFIC)) if pmc else None
            vpath = (valu_path or (PMC_VALU_LARGE if args.large else PMC_VALU)) if pmc else None
            traffic = None if tpath is None else pmc_traffic(names[dom], tpath)
            common = {"kernel": names[dom], "ms_per_launch": round(ms, 4), "ms_per_launch_is": TSTAT + " of the timed launches",
                      "launches_timed": tm[dom][0], "traffic": traffic,
                      "traffic_kind": TRAFFIC_KIND,
                      "traffic_source": None if traffic is None else f"committed PMC passes {os.path.relpath(tpath, ROOT)}, kernel {names[dom]}",
                      "other_kernels_ms": {k: round(v[1], 4) for k, v in tm.items() if k != dom}, "timed_with": graphed_note}
            valu = None if vpath is None else pmc_valu(names[dom], vpath)
            out = dict(common, **mfma, valu_pmc=valu)
            if fp16_pipe and valu is not None and valu.get("valu_insts_per_launch"):
                # NAMED SECONDARY (not the roofline): what holds the split-fp16 kernels is VALU issue, not the matrix pipe.  Vector
                # wave-instructions per launch (a property of the code and the shapes; counted by the committed PMC pass named in
                # `inputs`) over the duration measured LIVE in this run, against 1024 SIMDs x 2.4 GHz / 4 cycles per wave64
                # instruction.  It says how busy the vector pipe is with the instructions THIS kernel chose to issue -- a kernel
                # that issued more would score higher -- so `frac` above stays the algorithmic figure of SURVEY.md 8(d).
                rate = valu["valu_insts_per_launch"] / (ms * 1e-3) / 1e9
                out["valu_issue_frac"] = round(rate / 614.4, 4)
                out["valu_issue"] = {"achieved": round(rate, 1), "peak": 614.4, "unit": "G vector wave-instructions/s",
                                     "inputs": {"valu_insts_per_launch": valu["valu_insts_per_launch"], "valu_insts_source": valu["source"],
                                                "ms_per_launch": round(ms, 4), "ms_source": "HIP events in this run",
                                                "peak": "1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction (transcendentals take two slots)"}}
                if valu.get("mfma_insts_per_launch") and valu.get("kernel_cycles_per_launch"):
                    # Round 6 (tools/ubench/mfma_valu_overlap3.hip, profiles/r06_mfma_valu_overlap.txt): on one SIMD vector and matrix
                    # instructions do not run beside each other -- an MFMA 16x16x32 holds the vector issue for 11-13 of its 16 cycles,
                    # from either wave of the SIMD -- so a vector-heavy kernel's floor is the SUM of its vector issue cycles and its
                    # MFMAs' hold.  All three inputs are the committed PMC pass's (instructions are a property of the code, the cycle
                    # count is that pass's own): nothing here is measured live.
                    vc, mc = valu["valu_insts_per_launch"] * 4.0 * 1.25, valu["mfma_insts_per_launch"] * 11.5
                    simd = valu["kernel_cycles_per_launch"] * 1024.0
                    out["issue_bound"] = {"frac": round((vc + mc) / simd, 3), "vector_issue_cycles": vc, "mfma_hold_cycles": mc, "simd_cycles": simd,
                                          "note": "named secondary: (vector wave-instructions x 4 cycles x 1.25 [a quarter of the loop's vector "
                                                  "instructions -- exp2, sqrt, packed fp32 -- take 8] + MFMAs x 11.5 cycles of held vector issue) / "
                                                  "(kernel cycles x 1024 SIMDs): the share of the kernel's SIMD cycles in which an instruction of it "
                                                  "can issue at all; DESIGN.md section 4 'Round 6: matrix and vector instructions on one SIMD'"}
            return out

        ev_note = ("HIP events around eager launches of the same step right after the timed region (a graph replay cannot carry events)"
                   if graphed else "HIP events inside the timed region")
        roofline = attention_roofline(timers, k16 if split else k32, split, ev_note,
                                      None if split else PMC_VALU_FP32, None if split else PMC_TRAFFIC_FP32,
                                      timed_steps=(eager_timed_steps if graphed else args.steps), live=pairs if split else None,
                                      pmc=not args.mixed and args.pixel_positions == 0)
        if args.mixed:
            roofline["note"] = ("mixed-size stream: FLOP and duration are means over the stream's layouts (each run once, eagerly, after the timed "
                                "region); no PMC pass is committed for this workload (traffic / valu_pmc null)")
        if strict is not None:
            strict["roofline"] = attention_roofline(strict.pop("_timers"), k32, False,
                                                    "HIP events around eager launches of the fp32 step right after its timed region",
                                                    PMC_VALU_FP32, PMC_TRAFFIC_FP32, timed_steps=3)
        result = {
            "metric": "slides/sec (DGDM fwd+bwd, 10k-node/768-feat graphs)", "value": round(world * args.batch * args.steps / dt, 3),
            "unit": "slides/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (" + ("attention products: every operand incl. P and dS as fp16 hi+lo pairs (~21 significand bits), fp32 accumulate; " if split else "attention: fp32 MFMA; ") +
                     {"f16x2": "dense layers: fp16 hi+lo operands with per-operand power-of-two scale, 3 MFMAs per product, fp32 accumulate",
                      "bf16x3": "dense layers: exact 3-way bf16 split, 6 MFMAs per product, fp32 accumulate",
                      "fp32": "dense layers: fp32 MFMA"}[ops.GEMM_MATH] +
                     (arithmetic_error_note() if split and ops.GEMM_MATH == "f16x2" else "") + ")",
            "data": "synthetic",
            "config": {"workload": (f"MIXED-SIZE STREAM (configs[4]): DGDM-Base pretrain_step fwd+bwd+AdamW, batch={args.batch} graphs of "
                                    f"1k..10k nodes (E = 5 N) per GPU, 8 batches cycled, feat={FEATS}, edge_attr=32, T=10, heads=8, "
                                    if args.mixed else
                                    f"DGDM-{'Large (configs[3])' if args.large else 'Base'} pretrain_step fwd+bwd+AdamW, batch={args.batch} x {args.nodes}-node/{args.edges}-edge "
                                    f"graphs per GPU, feat={FEATS}, edge_attr=32, T={cfg['num_diffusion_steps']}, heads={cfg['attention_heads']}, ") +
                                   (f"positions = raster grid at {args.pixel_positions:g}-pixel pitch (NOT the headline's U[0,1)^2), "
                                    if args.pixel_positions > 0 else "") +
                                   f"{'eval (dropout off)' if args.eval_mode else 'training mode (dropout 0.1)'}",
                       "global_batch": world * args.batch, "parallelism": f"dp{world}", "final_loss": round(loss_val, 5),
                       "launch": (graph_note or "HIP graph replay (training.GraphedPretrainStep)") if graphed else (graph_note or "eager"),
                       "input_copy": ("excluded: the batch is resident in the recording's own input buffers (GraphedPretrainStep.input_buffers) "
                                      "before the timed region; rounds 1-3 copied it device-to-device every step (130 MB, ~0.09 ms)"
                                      if graphed and stream is None else
                                      "device-to-device copy of each batch into its recording's buffers, inside the timed region" if graphed
                                      else "none (eager step reads the resident batch)")},
            "roofline": roofline,
        }
        if balance_note is not None:
            result["config"].update(balance_note)
        if per_rank is not None:
            result["per_rank"] = per_rank
        if reducer is not None and reducer.flat is not None:
            result["gradient_exchange"] = {"collective": "all_reduce(AVG) of one flat fp32 buffer (RCCL), live parameters only",
                                           "bytes": reducer.nbytes, "bucket_bytes": reducer.bucket_nbytes,
                                           "messages_per_step": 1 if graphed else 2, "early_launches": reducer.stats.get("early_launches", 0)}
        if raster is not None:
            result["raster_positions"] = raster
        if strict is not None:
            result["strict_fp32"] = strict
        if sustained is not None:
            result["sustained"] = sustained
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if not args.no_gather:
            result["gather_roofline"] = gather_microbench(dev)
            result["projection_roofline"] = projection_microbench(dev)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args.nodes, args.edges)
        if world == 1 and not args.no_sample_loop and not args.mixed and not args.large:
            result["sample_loop"] = sample_loop_bench(model, dev)
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
