#!/usr/bin/env python3
"""Headline benchmark: generated mel-frames/s (+ RTF) of 32-NFE F5-TTS-Base with CFG 2.0 and the Vocos vocoder on
10 s synthetic utterances (BASELINE.json configs[1]); one utterance per GPU per step (weak scaling).

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

A step = one pass of the hot path over one utterance per rank: RCCL broadcast of the reference-audio latents from
rank 0 (N > 1 only), f5hip_cfm_sample (32 Euler steps x (cond + uncond) DiT forwards), ref-frame strip,
f5hip_vocos_decode, D2H of the waveform.  Inputs (cond mel, token ids, noise) are resident in HBM before the timed
region.  Weights/inputs are seeded synthetic (tts-indic-server-f5_amd/synth.py): no checkpoints offline.
Prints ONE JSON line (rank 0)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

N_REF, N_TOTAL, N_REF_IDS, N_GEN_IDS, STEPS_NFE, CFG, SWAY = 468, 1404, 60, 120, 32, 2.0, -1.0
PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md


def gemm_algorithmic_flops(n=N_TOTAL, dim=1024, depth=22, ff_mult=2, mel=100, nfe=STEPS_NFE, branches=2):
    """FLOPs the GEMM kernel class must do per utterance inside the ODE loop (SURVEY §8(d) per-token MACs, real
    dims, no padding, the split-bf16 x3 NOT counted, step-invariant work hoisted out NOT counted)."""
    per_tok = depth * (4 * dim * dim + 2 * ff_mult * dim * dim)      # qkv + out + ff1 + ff2 = 184.55 M at Base
    per_tok += mel * dim                                              # x part of the input projection
    per_tok += 2 * (dim // 16) * 31 * dim                             # conv_pos_embed, 2 grouped convs
    per_tok += dim * mel                                              # proj_out
    return 2.0 * per_tok * n * branches * nfe


def attn_algorithmic_flops(n=N_TOTAL, dim=1024, depth=22, nfe=STEPS_NFE, branches=2):
    return 2.0 * depth * 2 * n * dim * n * branches * nfe


def cpu_baseline(sd, vsd, cond, text, y0, n_threads):
    """The oracle (a port of the reference's fp32 CPU path) on the host cores, bounded sample: 2 Euler steps with CFG
    (4 DiT forwards at N = 1404) + one Vocos decode; the ODE part is scaled x16 to the 32-step job."""
    from oracle import dit_oracle as O
    from oracle import vocos_oracle as V
    torch.set_num_threads(n_threads)
    t0 = time.time()
    out, _ = O.cfm_sample(sd, O.F5_BASE, cond, text, N_TOTAL, steps=2, cfg_strength=CFG, sway_sampling_coef=SWAY, y0=y0,
                          keep_trajectory=False)
    t_ode = time.time() - t0
    t0 = time.time()
    V.vocos_decode(vsd, out[:, N_REF:].permute(0, 2, 1))
    t_voc = time.time() - t0
    wall = t_ode * (STEPS_NFE / 2) + t_voc
    return {"value": round((N_TOTAL - N_REF) / wall, 3), "unit": "mel-frames/s", "cores": n_threads, "kind": "port",
            "sample": f"2 of 32 Euler steps with CFG (4 DiT forwards, N=1404) = {t_ode:.1f} s scaled x16, + 1 Vocos decode = {t_voc:.2f} s",
            "rtf": round(wall / ((N_TOTAL - N_REF - 1) * 256 / 24000.0), 3)}


PMC_TRAFFIC_BYTES_PER_LAUNCH = 79.6e6   # 67.2 MB fetch + 12.4 MB write per launch (algorithmic minimum: 30-39 MB)

GEMM_MODES = {
    1: "plain bf16 everywhere (misses the 1e-3 mel bound)",
    2: "bf16x3 split everywhere (hi*hi+hi*lo+lo*hi, fp32 acc) - strict parity mode, 1.1e-4 mel RMS",
    3: "mixed parity mode: fp16 x fp16 (fp32 acc) for the transformer-block GEMMs, bf16x3 split for the GEMMs on the ODE state / embeddings - 3e-4 mel RMS vs the 1e-3 bound",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gemm-planes", type=int, default=3, choices=[1, 2, 3],
                    help="3 = mixed parity mode (fp16 block GEMMs + bf16x3 state GEMMs, default), 2 = bf16x3 everywhere, 1 = plain bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=1, help="utterances per GPU per step (1 = BASELINE configs[1]; 8 = configs[2] per-GPU share)")
    ap.add_argument("--vocoder", default="vocos", choices=["vocos", "bigvgan"], help="bigvgan = BASELINE configs[3]")
    args = ap.parse_args()

    if os.environ.get("F5HIP_BENCH_WATCHDOG"):          # dump every thread's stack and exit if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["F5HIP_BENCH_WATCHDOG"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # one process per GPU; F5HIP_DIST_BACKEND=gloo + several ranks on one card is only for rehearsing the N > 1 path on a 1-GPU box
    backend = os.environ.get("F5HIP_DIST_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count())
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{dev_index}"))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    torch.cuda.set_device(dev_index)
    dev = torch.device(f"cuda:{dev_index}")

    from tts_indic_server_f5_amd import _lib, synth
    from tts_indic_server_f5_amd.model import F5TTS_BASE, F5HipModel
    from tts_indic_server_f5_amd.sharding import broadcast_ref_latents
    from tts_indic_server_f5_amd.vocoder import F5HipVocos

    sd, vsd = synth.dit_state_dict(), synth.vocos_state_dict()
    model = F5HipModel(F5TTS_BASE, sd, gemm_planes=args.gemm_planes, device=dev)
    vocos = F5HipVocos(vsd, gemm_planes=min(args.gemm_planes, 2), device=dev)
    bigv = None
    if args.vocoder == "bigvgan":
        from tts_indic_server_f5_amd.vocoder import F5HipBigVGAN
        bigv = F5HipBigVGAN(synth.bigvgan_state_dict(), gemm_planes=min(args.gemm_planes, 2), device=dev)
    B = args.batch

    # rank 0 owns the reference-audio latents; every rank has its own gen text + noise (distinct seeds)
    gc = torch.Generator().manual_seed(14)
    cond0 = torch.randn(N_REF + 1, 100, generator=gc).to(dev) if rank == 0 else None
    ids = synth.text_ids(N_REF_IDS, N_GEN_IDS, seed=synth.SEED_TEXT + rank)[0]
    ref_ids0 = synth.text_ids(N_REF_IDS, 0)[0].to(dev) if rank == 0 else None
    gen_ids = ids[N_REF_IDS:].to(dev)
    y0 = torch.stack([synth.noise(N_TOTAL, rank * B + i) for i in range(B)]).to(dev)

    def one_step(exchange=True):
        # exchange=False (rank 0's untimed profiling pass) must not enter a collective the other ranks never join
        cond, ref_ids = broadcast_ref_latents(cond0, ref_ids0, dev) if exchange else (cond0, ref_ids0)
        text = torch.cat([ref_ids, gen_ids])[None].expand(B, -1)
        out, _ = model.sample(cond[None].expand(B, -1, -1), text, N_TOTAL, steps=STEPS_NFE, cfg_strength=CFG, sway_sampling_coef=SWAY, y0=y0)
        mel = out[:, N_REF:, :].permute(0, 2, 1)
        wave = bigv(mel) if bigv is not None else vocos.decode(mel)
        return wave.reshape(B, -1).cpu()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wave = one_step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    gen_frames = N_TOTAL - N_REF
    value = gen_frames * B * world * args.steps / dt
    audio_s = wave.numel() / 24000.0

    result = None
    if rank == 0:
        # per-kernel-class durations: one more pass with HIP events around every launch on the launch stream
        L = _lib.lib()
        L.f5hip_set_profiling(1)
        one_step(exchange=False)
        torch.cuda.synchronize()
        import ctypes as C
        prof = {}
        for cls in ("gemm", "attn", "ln", "other", "vocos"):
            ms, n = C.c_double(0), C.c_int64(0)
            L.f5hip_get_profile(cls.encode(), C.byref(ms), C.byref(n))
            prof[cls] = {"total_ms": round(ms.value, 3), "launches": n.value}
        L.f5hip_set_profiling(0)
        g = prof["gemm"]
        # the profiled pass also ran the hoisted / Vocos GEMMs; their share of launches and time is < 2 %
        gemm_flops = gemm_algorithmic_flops() * B
        achieved = gemm_flops / (g["total_ms"] * 1e-3) / 1e12 if g["total_ms"] > 0 else 0.0
        att = prof["attn"]
        attn_tf = attn_algorithmic_flops() * B / (att["total_ms"] * 1e-3) / 1e12 if att["total_ms"] > 0 else 0.0
        result = {
            "metric": "mel-frames/sec + RTF, F5-TTS-Base 32-NFE, 10s utterance", "value": round(value, 1),
            "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "fp16/bf16 MFMA, fp32 accumulate" if args.gemm_planes == 3 else "bf16", "data": "synthetic",
            "rtf": round(dt / args.steps / audio_s, 6),
            "config": {"workload": f"F5-TTS-Base, 32 NFE + CFG=2.0 + sway -1, {'BigVGAN' if bigv is not None else 'Vocos'}, {B} x 10 s utterance (N=1404, 936 generated frames) per GPU per step",
                       "gemm_mode": GEMM_MODES[args.gemm_planes],
                       "attention": "bf16 MFMA, fp32 softmax", "parallelism": f"utterance-sharded x{world}, RCCL broadcast of ref latents"},
            "roofline": {"bound": "mfma", "kernel": "gemm3_kernel / gemm_kernel (all GEMM instantiations)", "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
                         # memory-side bytes per GEMM launch from the PMC passes committed in profiles/ (not collected live: rocprofv3 only)
                         "traffic": PMC_TRAFFIC_BYTES_PER_LAUNCH if (args.gemm_planes == 3 and B == 1) else None,
                         "traffic_source": "profiles/r01_pmc_hbm_traffic.txt: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate passes, mean over the fp16 block GEMMs",
                         "avg_launch_us": round(g["total_ms"] * 1e3 / max(1, g["launches"]), 2), "launches": g["launches"],
                         "executed_mfma_x": {1: 1, 2: 3, 3: 1.02}[args.gemm_planes],
                         "attn_tflops": round(attn_tf, 1), "attn_frac": round(attn_tf / PEAK_BF16_TFLOPS, 4)},
            "kernel_ms": prof,
        }
        if not args.no_cpu_baseline and world == 1 and B == 1 and bigv is None:
            n_threads = min(len(os.sched_getaffinity(0)), 32)
            result["cpu_baseline"] = cpu_baseline(sd, vsd, cond0.cpu()[None], torch.cat([ref_ids0, gen_ids]).cpu()[None], y0.cpu(), n_threads)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
