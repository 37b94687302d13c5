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


def flops_per_pair(cfg, B_local, mean_words=None, mean_tokens=None):
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
    return 3 * (vit + cfg.top_k * expert + local) + txt


def traffic_from_profile(config, gb, world):
    """HBM bytes per launch of the dominant kernel from the COMMITTED PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate passes, FETCH doubled per the gfx950 note; tools/collect_traffic.py).  Not measured by this run: reported next
    to the file it comes from, and only for the exact workload it was collected on."""
    for name in ("r02_traffic_c.json", "r02_traffic_b.json", "r02_traffic.json", "r01_traffic.json"):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
            e = d[f"{config}_gb{gb}_n{world}"]
            return {"bytes_per_launch": e["hbm_bytes_per_launch"], "file": f"profiles/{name}", "kernel": e.get("kernel", "medmoe_gemm_nt launches")}
        except Exception:
            continue
    return None


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
    cfg = config_by_name(args.config)
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
    prof = []
    sync()
    t0 = time.perf_counter()
    for it in range(args.steps):
        # HIP events around every gemm_nt launch (same stream), on the FIRST timed step of rank 0 only: 1600 event
        # records per step cost ~7 ms of host time, which at small per-rank batches would make rank 0 the straggler
        ops.PROFILE = prof if (rank == 0 and it == 0) else None
        out = eng.train_step(batch)
    sync()
    dt = time.perf_counter() - t0
    ops.PROFILE = None
    tmax = torch.tensor([dt], device=eng.device)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())
    loss = float(out["loss"])

    if rank == 0:
        pairs_per_s = gb * args.steps / dt
        mean_words = float(eng.cap_lens.float().clamp(max=cfg.max_len).mean())      # words per caption of this batch (device -> host, after timing)
        mean_tokens = float(batch["attn_mask"].float().sum(1).mean()) if getattr(eng, "text_varlen", False) else None
        fpp = flops_per_pair(cfg, B, mean_words, mean_tokens)
        fpp_max = flops_per_pair(cfg, B)
        step_tflops = pairs_per_s * fpp / 1e12 / world
        gemm_ms = sum(p[1].elapsed_time(p[2]) for p in prof)
        gemm_flops = sum(p[0] for p in prof)
        gemm_tf = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        res = {
            "metric": "image-text pairs/sec at global batch 1024" if gb == 1024 else f"image-text pairs/sec at global batch {gb}",
            "value": pairs_per_s, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.config}: ViT-{ {768: 'B', 1024: 'L'}.get(cfg.d_v, cfg.d_v) }/{cfg.patch} + {cfg.n_layer_t}-layer text tower (frozen), "
                                   f"{cfg.n_expert} experts top-{cfg.top_k}, {cfg.img_size}x{cfg.img_size}x3 + {cfg.max_len} tokens, fwd+bwd+clip+Adam",
                       "global_batch": gb, "per_gpu_batch": B, "parallelism": f"dp{world}", "loss": loss,
                       "hbm_peak_gb": torch.cuda.max_memory_allocated() / 1e9},
            "roofline": {"bound": "mfma",
                         "kernel": "every medmoe_gemm_nt launch of the step: gemm_nt4w_kernel (plain + GROUPED builds) and, for narrow / short / "
                                   "odd shapes, gemm_nt256_kernel / gemm_nt_kernel",
                         "achieved": gemm_tf, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": gemm_tf / PEAK_BF16_TFLOPS,
                         "traffic": None, "traffic_from_profile": traffic_from_profile(args.config, gb, world),
                         "launches": len(prof), "avg_launch_ms": gemm_ms / max(1, len(prof)),
                         "gemm_share_of_step": gemm_ms / (dt / args.steps * 1e3) if dt > 0 else None,
                         "events": "HIP events on the launch stream around every launch of the first timed step",
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
        if not args.no_cpu_baseline and world == 1:
            del eng
            torch.cuda.empty_cache()
            res["cpu_baseline"] = cpu_baseline(args.config)
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
