#!/usr/bin/env python3
"""Throughput benchmark of the MedMoE contrastive training step on MI355X.

  python bench.py --gpus N --steps K --warmup W [--config cfg2|cfg1|cfg0|cfg3|cfg4|tiny] [--global-batch G]

One process per GPU (for N>1 launch through torch.distributed.run; RANK/LOCAL_RANK/WORLD_SIZE
are read from the env).  A "step" is one full optimisation step of the hot path on one synthetic
batch already resident in HBM: ViT + frozen text tower + MoE forward, GLoRIA local/global +
router-CE losses, full backward, gradient all-reduce (N>1), global-norm clip + Adam.
Default workload = the configuration BASELINE.json's metric is quoted on: ViT-B/16 + 12-layer
text tower, 8 experts top-2 (configs[2]) at GLOBAL batch 1024, split evenly over the N ranks
(strong scaling; per-rank batch 1024/N).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md


def flops_per_pair(cfg, B_local, mean_words=None, mean_tokens=None, train_text=False):
    """Algorithmic training FLOPs per pair, text tower frozen (SURVEY.md section 8d formulas).  The local loss costs
    4 * P * D flop per (image, caption WORD): the reference slices every caption to its own length
    (losses.py:985 `words_emb[i, :, :words_num]`), so the count uses the batch's mean caption length `mean_words`;
    mean_words=None gives the SURVEY table's upper bound (every caption max_len words long).  mean_tokens: the text tower counted on
    the non-padding tokens only (what the build computes since the tower runs on packed tokens; the reference runs all max_len positions,
    which is the mean_tokens=None figure) - attention per caption taken at the mean length, a lower bound of the mean of squares."""
    N, Dv, L, ff, P = cfg.n_tok_v, cfg.d_v, cfg.n_layer_v, cfg.ff_v, cfg.n_patch
    vit = L * (2 * N * Dv * 3 * Dv + 4 * N * N * Dv + 2 * N * Dv * Dv + 4 * N * Dv * ff) + 2 * P * (3 * cfg.patch ** 2) * Dv
    T, D, Lt, fft = cfg.max_len, cfg.d_t, cfg.n_layer_t, cfg.ff_t
    Tt = T if mean_tokens is None else mean_tokens
    txt = Lt * (2 * Tt * D * 3 * D + 4 * Tt * Tt * D + 2 * Tt * D * D + 4 * Tt * D * fft)
    Do = cfg.d_out
    expert = 4 * 2 * P * Dv * Do + P * 4 * 2 * (Do * (Do // 2) + Do // 2)
    local = B_local * 4 * P * Do * (T if mean_words is None else mean_words)
    return 3 * (vit + cfg.top_k * expert + local) + (3 if train_text else 1) * txt


def traffic_from_profile(config, gb, world):
    """HBM bytes per launch of the dominant kernel from the COMMITTED PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate passes, FETCH doubled per the gfx950 note of MI355X_MICROARCH.md; tools/collect_traffic.py).  Not measured by this run:
    reported with the file it comes from, and only for the exact workload it was collected on."""
    for name in ("r03_traffic.json", "r02_traffic_c.json", "r02_traffic_b.json", "r02_traffic.json", "r01_traffic.json"):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
            e = d[f"{config}_gb{gb}_n{world}"]
            return {"bytes_per_launch": e["hbm_bytes_per_launch"], "file": f"profiles/{name}", "kernel": e.get("kernel", "medmoe_gemm_nt launches"),
                    "algorithmic_bytes_per_launch": e.get("algorithmic_bytes_per_launch")}
        except Exception:
            continue
    return None


PEAK_HBM_GBS = 8000.0          # HBM3E peak (spec), MI355X_MICROARCH.md; 6.3 TB/s is what a streaming copy achieves


def kernel_table(prof, top=6):
    """Per-kernel totals of one profiled step: [(label, launches, ms, work, unit)] sorted by time, and the `roofline.kernels` list."""
    agg = {}
    for label, work, unit, ev0, ev1, _ in prof:
        a = agg.setdefault(label, [0, 0.0, 0.0, unit, True])
        a[0] += 1; a[1] += ev0.elapsed_time(ev1)
        if work is None:
            a[4] = False
        else:
            a[2] += work
            a[3] = unit
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    out = []
    for label, (n, ms, work, unit, complete) in rows[:top]:
        e = {"name": label, "launches": n, "ms_per_step": ms}
        if complete and unit == "flop" and ms > 0:
            tf = work / (ms * 1e-3) / 1e12
            e.update(achieved=tf, unit="TFLOP/s", bound="mfma", frac=tf / PEAK_BF16_TFLOPS)
        elif complete and unit == "byte" and ms > 0:
            gbs = work / (ms * 1e-3) / 1e9
            e.update(achieved=gbs, unit="GB/s", bound="hbm", frac=gbs / PEAK_HBM_GBS)
        out.append(e)
    return rows, out


def synthetic_batch(cfg, B, seed, device):
    """SURVEY 8d synthetic inputs, generated on the device (nothing crosses PCIe in the timed region)."""
    g = torch.Generator(device=device).manual_seed(seed)
    T = cfg.max_len
    img = torch.randn(B, 3, cfg.img_size, cfg.img_size, generator=g, device=device).to(torch.bfloat16)
    lens = torch.randint(min(8, T), T + 1, (B,), generator=g, device=device)
    ids = torch.randint(3, cfg.vocab, (B, T), generator=g, device=device)
    pos = torch.arange(T, device=device)[None]
    ids[:, 0] = 1
    ids = torch.where(pos == (lens[:, None] - 1), torch.full_like(ids, 2), ids)
    ids = torch.where(pos >= lens[:, None], torch.zeros_like(ids), ids)
    return {"image": img, "ids": ids, "attn_mask": (pos < lens[:, None]).long(), "token_type": torch.zeros_like(ids),
            "label": torch.randint(0, cfg.n_expert, (B,), generator=g, device=device)}


def cap_lens_of(batch, cfg):
    """Mean words per caption as the engine counts them (cap_len = words + 1, clamped to max_len) for the synthetic ids: every
    non-padding token except [CLS] / [SEP] is a word."""
    n_tok = batch["attn_mask"].float().sum(1)
    return float((n_tok - 2 + 1).clamp(min=1, max=cfg.max_len).mean())


def bench_module_path(args, cfg, batch, world, sync):
    """`src/train.py experiment=pretraining_medmoe_<config>` as far as the model node: MedMoEPretrainingLightningModule built by the Hydra
    composer with model.fused_step=true, stepped through `training_step` on the same synthetic batch."""
    os.environ.setdefault("PROJECT_ROOT", ROOT)
    from medmoe_amd.hydra_lite import compose, instantiate
    hc = compose(os.path.join(ROOT, "configs"), "train.yaml", [f"experiment=pretraining_medmoe_{args.config}"])
    lit = instantiate(hc.model)
    if not lit.fused_step:
        raise RuntimeError("experiment does not switch model.fused_step on")
    ec = lit.model.engine.cfg
    for k in ("d_v", "n_layer_v", "n_expert", "top_k", "max_len", "img_size", "patch", "expert_fp8"):
        if getattr(ec, k) != getattr(cfg, k):
            raise RuntimeError(f"experiment geometry differs from --config {args.config}: {k}")
    lit.configure_optimizers()
    lit.configure_fused(hc.trainer.accumulate_grad_batches, hc.trainer.gradient_clip_val)
    mb = {"image": batch["image"], "label": batch["label"], "caption": {"ids": batch["ids"], "attn_mask": batch["attn_mask"],
                                                                         "token_type": batch["token_type"]}}
    for i in range(args.warmup):
        lit.training_step(mb, i)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = lit.training_step(mb, i)
    sync()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=batch["image"].device)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t)
    gb = batch["image"].shape[0] * world
    return {"what": f"python src/train.py experiment=pretraining_medmoe_{args.config}: MedMoEPretrainingLightningModule.training_step in its fused "
                    "mode (model.fused_step: true) - Engine.train_step behind the reference's module interface",
            "ms_per_step": dt / args.steps * 1e3, "value": gb * args.steps / dt, "unit": "pairs/s", "loss": float(loss)}


def bench_ref_swin(args, world, local_rank):
    """--config ref_swin: the REFERENCE's own model (configs/experiment/pretraining_medmoe.yaml + configs/model/med-moe.yaml: HF Swin-T tower, six
    pyramid experts over its four stages, 56 x 56 = 3136 local regions, frozen BERT-geometry text tower, captions of 25 tokens, per-device batch
    32) behind the reference-named module.  `value` is the fused step (model.fused_step=true -> medmoe_amd.swin_engine.SwinEngine: hand-scheduled
    launches, fused clip + Adam); `module_path` is the same model through torch autograd over the HIP kernels + src.losses + clip_grad_norm_ +
    torch Adam (fused), what the reference experiment's default config runs.  --path engine | module | both."""
    os.environ.setdefault("PROJECT_ROOT", ROOT)
    from medmoe_amd.hydra_lite import compose, instantiate
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:                                   # data parallel: the fused step only (embedding all-gather, arena all-reduce under the tower's backward)
        args.path = "engine"
    if args.global_batch and args.global_batch % world:
        raise SystemExit("global batch must divide evenly over the ranks")
    B = (args.global_batch // world) if args.global_batch else 32          # per rank; default = the reference's per-device batch (weak scaling)
    base = ["experiment=pretraining_medmoe", "model.model.vision.arch=swin_t"]

    def build(fused):
        hc = compose(os.path.join(ROOT, "configs"), "train.yaml", base + (["model.fused_step=true"] if fused else []))
        lit = instantiate(hc.model)
        lit.train()
        cfg = lit.model.cfg
        b = synthetic_batch(cfg, B, 12345 + rank, lit.model.device)
        b["label"] = b["label"] % int(hc.model.model.vision.num_experts)
        mb = {"image": b["image"], "label": b["label"], "caption": {"ids": b["ids"], "attn_mask": b["attn_mask"], "token_type": b["token_type"]}}
        return hc, lit, mb

    def timed(step):
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        dt = time.perf_counter() - t0
        if world > 1:                               # the slowest rank's clock
            t = torch.tensor([dt], device="cuda", dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t)
        return dt / args.steps, float(loss.detach())

    res_mod = None
    if args.path in ("module", "both"):
        hc, lit, mb = build(False)
        params = [p for p in lit.parameters() if p.requires_grad]
        opt = torch.optim.Adam(params, lr=float(hc.model.optimizer.lr), weight_decay=float(hc.model.optimizer.weight_decay), fused=True)
        clip = float(hc.trainer.gradient_clip_val)

        def step_module():
            opt.zero_grad()
            loss = lit.training_step(mb, 0)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(params, clip)
            opt.step()
            return loss
        dt, loss = timed(step_module)
        res_mod = {"ms_per_step": dt * 1e3, "value": B / dt, "loss": loss,
                   "what": "torch autograd over the HIP kernels + src.losses + clip_grad_norm_ + torch.optim.Adam(fused=True)"}
        n_par = sum(p.numel() for p in params) / 1e6
        cfg = lit.model.cfg
        del lit, opt, params
        torch.cuda.empty_cache()
    if args.path in ("engine", "both"):
        hc, lit, mb = build(True)
        lit.configure_optimizers()
        lit.configure_fused(1, float(hc.trainer.gradient_clip_val))
        n_par = sum(p.numel() for p in lit.parameters() if p.requires_grad) / 1e6
        dt, loss = timed(lambda: lit.training_step(mb, 0))
        cfg = lit.model.cfg
    res = {"metric": f"image-text pairs/sec at global batch {B * world}", "value": B * world / dt, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "strong" if args.global_batch else "weak", "vs_baseline": None,
           "dtype": "bf16", "data": "synthetic",
           "config": {"workload": "ref_swin: the reference's own model - HF Swin-T tower + 6 pyramid experts (3136 local regions) + frozen 12-layer text tower, "
                                  f"224x224x3 + {cfg.max_len} tokens, " + ("fused step (SwinEngine: hand-scheduled launches, clip 0.25, fused Adam)"
                                                                           if args.path != "module" else "module path (torch autograd, torch fused Adam)"),
                       "global_batch": B * world, "per_gpu_batch": B, "parallelism": f"dp{world}", "loss": loss,
                       "trainable_parameters_m": n_par, "hbm_peak_gb": torch.cuda.max_memory_allocated() / 1e9},
           "roofline": None, "note": "secondary workload (BASELINE.md section 4); the BASELINE.json configs name ViT towers"}
    if res_mod is not None and args.path == "both":
        res["module_path"] = res_mod
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def usable_cores() -> int:
    """Threads this process may actually run on: cgroup CPU quota (the GPU box gives a one-GPU job a share of
    the host, far fewer than os.cpu_count()), then the affinity mask, then cpu_count.  MEDMOE_CPU_THREADS overrides."""
    if os.environ.get("MEDMOE_CPU_THREADS"):
        return max(1, int(os.environ["MEDMOE_CPU_THREADS"]))
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except Exception:
            pass
    return min(n, 64)          # beyond ~64 threads the fp32 oracle step stops scaling (small GEMMs)


def _cpu_steps(cfg_name, pairs, warmup, steps, cores):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import medmoe_oracle as O
    ocfg = O.config_by_name(cfg_name)
    torch.set_num_threads(cores)
    p = O.init_params(ocfg, seed=0)
    train = [v.requires_grad_(True) for k, v in p.items() if not k.startswith("text.")]
    opt = torch.optim.Adam(train, lr=5e-5)
    batch = O.synthetic_batch(ocfg, pairs)
    vocab = O.Vocab.synthetic(ocfg.vocab)
    times = []
    for it in range(warmup + steps):
        t0 = time.perf_counter()
        opt.zero_grad()
        out = O.model_step(batch, p, ocfg, vocab)
        out["loss"].backward()
        torch.nn.utils.clip_grad_norm_(train, 0.25)
        opt.step()
        if it >= warmup:
            times.append(time.perf_counter() - t0)
    times.sort()
    return times[len(times) // 2], times[0]


def cpu_baseline(cfg_name):
    """The CPU oracle (a port of the reference path; the reference's Python cannot travel to this box) on this host's cores.
    `value` follows SURVEY.md 8(d): BASELINE configs[0] exactly (ViT-Ti/16 + 2-layer text tower, 2 experts top-1, 32 pairs,
    T = 25), fp32 fwd + bwd + clip + Adam, 2 warm-up steps, median of 10.  `same_model`: the benchmarked model itself at 8
    pairs per step (its local loss is O(B^2), so this is NOT the batch-1024 workload - a bounded sample of it)."""
    cores = usable_cores()
    med, best = _cpu_steps("cfg0", 32, 2, 10, cores)
    res = {"value": 32 / med, "unit": "pairs/s", "cores": cores, "kind": "port",
           "sample": "SURVEY 8(d) protocol: configs[0] (ViT-Ti/16 + 2-layer text tower, 2 experts top-1, T=25), 32 synthetic pairs/step, "
                     "fp32 fwd+bwd+clip+Adam, 2 warm-up + median of 10 steps"}
    if cfg_name in ("cfg1", "cfg2"):
        med2, _ = _cpu_steps(cfg_name, 8, 1, 3, cores)
        res["same_model"] = {"value": 8 / med2, "unit": "pairs/s",
                             "sample": f"{cfg_name} model, 8 synthetic pairs/step (not the benchmarked batch), 1 warm-up + median of 3 steps"}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--global-batch", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train-text", action="store_true",
                    help="cfg.freeze_text = False: the text tower trains too (reference freeze_bert: false; SURVEY 8(d) column 'train (full)') - "
                         "NOT the BASELINE metric's workload, which freezes it as the reference experiment does")
    ap.add_argument("--path", default="both", choices=["engine", "module", "both"],
                    help="engine: Engine.train_step only; module / both (default): also time the Hydra-built LightningModule's fused training_step "
                         "and report it as the secondary field `module_path`")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # MEDMOE_DIST_BACKEND=gloo rehearses the N>1 path with several ranks on ONE GPU (RCCL wants one device per rank)
    backend = os.environ.get("MEDMOE_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            torch.distributed.init_process_group(backend)

    from medmoe_amd import ops
    from medmoe_amd.config import config_by_name
    from medmoe_amd.engine import Engine
    if args.config == "ref_swin":
        return bench_ref_swin(args, world, local_rank)
    cfg = config_by_name(args.config)
    if args.train_text:
        cfg.freeze_text = False
    for kv in filter(None, os.environ.get("MEDMOE_OPTS", "").split(",")):      # measurement only: kernel-selection switches "key=value,..."
        ops.set_option(*(int(v) for v in kv.split("=")))
    gb = args.global_batch or {"cfg2": 1024, "cfg1": 256, "cfg0": 32, "cfg3": 64, "cfg4": 256, "tiny": 16}.get(args.config, 256)
    if gb % world:
        raise SystemExit("global batch must divide evenly over the ranks")
    B = gb // world
    eng = Engine(cfg, f"cuda:{local_rank}", seed=0)
    batch = synthetic_batch(cfg, B, 12345 + rank, eng.device)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.train_step(batch)
    sync()
    t0 = time.perf_counter()
    for it in range(args.steps):
        out = eng.train_step(batch)
    sync()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=eng.device)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())
    loss = float(out["loss"])

    # ONE more step, outside the timed region, with HIP events around every launch (on the stream the launch goes to): per-kernel
    # durations for the roofline fields.  Every rank runs it (the step holds collectives); rank 0 reports.
    prof = []
    mean_tokens = float(batch["attn_mask"].float().sum(1).mean()) if getattr(eng, "text_varlen", False) else None
    ops.ROWS_HINT = int(batch["attn_mask"].sum()) if mean_tokens is not None else 0
    ops.PROFILE = prof
    eng.train_step(batch)
    sync()
    ops.PROFILE = None

    # the drop-in path: the reference-named LightningModule built from the Hydra tree in its fused mode (model.fused_step: true), same
    # workload, same number of steps - a secondary field; `value` above stays the engine's own number
    module_path = None
    if args.path in ("module", "both") and args.config in ("cfg1", "cfg2", "cfg3", "cfg4") and cfg.freeze_text:
        try:
            del eng
            eng = None
            torch.cuda.empty_cache()
            module_path = bench_module_path(args, cfg, batch, world, sync)
        except Exception as e:                                       # the headline line must survive a failure here
            module_path = {"error": f"{type(e).__name__}: {e}"[:300]}

    if rank == 0:
        pairs_per_s = gb * args.steps / dt
        mean_words = float(cap_lens_of(batch, cfg))                  # words per caption of this batch (after timing)
        fpp = flops_per_pair(cfg, B, mean_words, mean_tokens, not cfg.freeze_text)
        fpp_max = flops_per_pair(cfg, B, train_text=not cfg.freeze_text)
        step_tflops = pairs_per_s * fpp / 1e12 / world
        rows, kernels = kernel_table(prof)
        dom = "gemm_nt4w_kernel"
        d_n, d_ms, d_flop, d_bytes = 0, 0.0, 0.0, 0.0
        fam_ms, fam_flop = 0.0, 0.0
        for label, work, unit, ev0, ev1, detail in prof:
            if label.startswith("gemm_nt") and not label.startswith("gemm_tn") and work is not None:
                fam_ms += ev0.elapsed_time(ev1); fam_flop += work
            if label == dom and work is not None:
                d_n += 1; d_ms += ev0.elapsed_time(ev1); d_flop += work
                if detail and detail[0] == "nt":
                    d_bytes += detail[5]
        gemm_tf = d_flop / (d_ms * 1e-3) / 1e12 if d_ms > 0 else 0.0
        fam_tf = fam_flop / (fam_ms * 1e-3) / 1e12 if fam_ms > 0 else 0.0
        tr = traffic_from_profile(args.config, gb, world)
        step_ms = dt / args.steps * 1e3
        res = {
            "metric": "image-text pairs/sec at global batch 1024" if gb == 1024 else f"image-text pairs/sec at global batch {gb}",
            "value": pairs_per_s, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": step_ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.config}: ViT-{ {768: 'B', 1024: 'L'}.get(cfg.d_v, cfg.d_v) }/{cfg.patch} + {cfg.n_layer_t}-layer text tower ({'frozen' if cfg.freeze_text else 'TRAINED - not the BASELINE workload'}), "
                                   f"{cfg.n_expert} experts top-{cfg.top_k}{' fp8 expert weights' if cfg.expert_fp8 else ''}, {cfg.img_size}x{cfg.img_size}x3 + {cfg.max_len} tokens, fwd+bwd+clip+Adam",
                       "global_batch": gb, "per_gpu_batch": B, "parallelism": f"dp{world}", "loss": loss,
                       "hbm_peak_gb": torch.cuda.max_memory_allocated() / 1e9},
            "roofline": {"bound": "mfma",
                         "kernel": "gemm_nt4w_kernel (the 256x256-tile, four-wave NT GEMM: every Linear forward / dgrad of the ViT tower and the "
                                   "packed text tower, the local-loss dC; plain build - the GROUPED expert build is listed separately)",
                         "achieved": gemm_tf, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": gemm_tf / PEAK_BF16_TFLOPS,
                         "traffic": tr["bytes_per_launch"] if tr else None,
                         "traffic_source": (f"{tr['file']} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on the committed tree, FETCH "
                                            f"doubled per the gfx950 note; kernel {tr['kernel']}; the algorithmic bytes of the same launches are "
                                            f"the next field)") if tr else None,
                         "algorithmic_bytes_per_launch": d_bytes / max(1, d_n),
                         "launches": d_n, "avg_launch_ms": d_ms / max(1, d_n), "share_of_step": d_ms / step_ms if step_ms > 0 else None,
                         "events": "HIP events on the launch stream around every launch of ONE extra step after the timed region "
                                   "(at per-rank batches <= 512 the second stream's kernels share the chip: durations include that sharing)",
                         "nt_family": {"kernels": "gemm_nt4w_kernel + GROUPED build + gemm_nt256_kernel + gemm_nt_kernel (every medmoe_gemm_nt launch)",
                                       "achieved": fam_tf, "frac": fam_tf / PEAK_BF16_TFLOPS, "ms_per_step": fam_ms},
                         "kernels": kernels,
                         "profiled_step_ms": sum(r[1][1] for r in rows),
                         "whole_step": {"algorithmic_gflop_per_pair": fpp / 1e9, "achieved": step_tflops,
                                        "frac": step_tflops / PEAK_BF16_TFLOPS, "mean_words_per_caption": mean_words,
                                        "mean_text_tokens_per_caption": mean_tokens,
                                        "algorithmic_gflop_per_pair_at_max_len": fpp_max / 1e9,
                                        "frac_at_max_len": pairs_per_s * fpp_max / 1e12 / world / PEAK_BF16_TFLOPS,
                                        "note": "local-loss flops counted at each caption's own length, as the reference computes them "
                                                "(losses.py:985), and the text tower at the captions' own token counts (it runs on the packed "
                                                "non-padding tokens; MEDMOE_TEXT_VARLEN=0 computes all positions); the *_at_max_len fields are "
                                                "the SURVEY 8d table's upper bound with every caption and every text position at max_len"}},
        }
        if module_path is not None:
            if "ms_per_step" in module_path:
                module_path["vs_engine"] = module_path["ms_per_step"] / step_ms
            res["module_path"] = module_path
        if not args.no_cpu_baseline and world == 1:
            eng = None
            torch.cuda.empty_cache()
            res["cpu_baseline"] = cpu_baseline(args.config)
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
