#!/usr/bin/env python3
"""Headline benchmark: generated mel-frames/s (+ RTF) of 32-NFE F5-TTS-Base with CFG 2.0 and the Vocos vocoder on
10 s synthetic utterances (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W [--mode weak|strong] [--batch B] [--total-batch T] [--ragged]
  python bench.py --arch e2 --nfe 64 --batch 8        BASELINE configs[4] per GPU: E2-TTS Base (UNetT), 64 NFE, 8 chunks of 2340 frames (20 s)

One process per GPU.  With --gpus N > 1 and no torchrun environment, this process only LAUNCHES N fresh children (one rank per
GPU, RCCL = backend "nccl") before anything here touches a GPU, and relays rank 0's JSON line; under
`python -m torch.distributed.run` the ranks come from the environment instead.

A step = one pass of the hot path over the rank's units: RCCL broadcast of the reference-audio latents from rank 0 (N > 1),
f5hip_cfm_sample (32 Euler steps x (cond + uncond) DiT forwards), ref-frame strip, vocoder, D2H of the waveforms (and, in strong
mode, the gather of the waveforms to rank 0).
  weak   (default): every rank has --batch utterances of its own per step (1 = configs[1] per GPU); value = N x batch units / time.
  strong: --total-batch units (64 = configs[2]) are dealt over the ranks by sharding.shard_units (longest-processing-time first,
          the inference analogue of the reference's frame-budget batch sampler) and the waveforms are gathered on rank 0.
Inputs (cond mel, token ids, noise) are resident in HBM before the timed region.  Weights / inputs are seeded synthetic
(tts-indic-server-f5_amd/synth.py): no checkpoints offline.  Prints ONE JSON line (rank 0)."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_REF, N_TOTAL, N_REF_IDS, N_GEN_IDS, STEPS_NFE, CFG, SWAY = 468, 1404, 60, 120, 32, 2.0, -1.0
PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
TRAFFIC_FILES = [os.path.join(ROOT, "profiles", f) for f in ("r03_pmc_hbm_traffic.json", "r02_pmc_hbm_traffic.json")]   # rocprofv3 --pmc passes (tools/pmc_traffic.py), newest first
E2_FRAMES = 2340   # one long-form chunk of configs[4]: 468 reference + 1872 generated frames (20 s)

GEMM_MODES = {
    1: "plain bf16 everywhere (misses the 1e-3 mel bound)",
    2: "bf16x3 split everywhere (hi*hi+hi*lo+lo*hi, fp32 acc) - strict parity mode, 1.1e-4 mel RMS",
    3: "mixed parity mode: fp16 x fp16 (fp32 acc) for the transformer-block GEMMs, bf16x3 split for the GEMMs on the ODE state / embeddings - 3e-4 mel RMS vs the 1e-3 bound",
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gemm-planes", type=int, default=3, choices=[1, 2, 3],
                    help="3 = mixed parity mode (fp16 block GEMMs + bf16x3 state GEMMs, default), 2 = bf16x3 everywhere, 1 = plain bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default="weak", choices=["weak", "strong"])
    ap.add_argument("--batch", type=int, default=1, help="weak mode: utterances per GPU per step (1 = BASELINE configs[1]; 8 = configs[2] per-GPU share)")
    ap.add_argument("--total-batch", type=int, default=64, help="strong mode: utterances per step over all GPUs (64 = BASELINE configs[2])")
    ap.add_argument("--ragged", action="store_true", help="generated lengths U(6 s, 14 s) instead of 10 s (SURVEY section 8(d))")
    ap.add_argument("--vocoder", default="vocos", choices=["vocos", "bigvgan"], help="bigvgan = BASELINE configs[3]")
    ap.add_argument("--vocoder-planes", type=int, default=2, choices=[1, 2, 3],
                    help="BigVGAN conv operand precision: 2 = split bf16 (parity mode, default), 3 = one fp16 plane (fast mode, outside the 1e-4 waveform bound)")
    ap.add_argument("--arch", default="f5", choices=["f5", "e2"], help="f5 = F5-TTS-Base (DiT, 10 s units); e2 = E2-TTS Base (UNetT, 20 s chunks of 2340 frames: configs[4])")
    ap.add_argument("--nfe", type=int, default=None, help="Euler steps (default 32; configs[4] uses 64)")
    ap.add_argument("--cpu-full", action="store_true", help="cpu_baseline: one full 32-step run instead of 4 steps scaled x8 (about a minute of CPU time)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launcher (no GPU call in this process)
def launch_ranks(n, argv, poll_s=0.2, grace_s=10.0):
    """Starts n fresh ranks of this script and relays rank 0's stdout.  All children are polled: when one exits non-zero the others get
    `grace_s` seconds to follow (they are usually blocked in a collective with the dead rank) and are then terminated, so a failing rank
    cannot hang the launcher.  Every rank's stderr goes to gpurun_out/bench_rank<r>.err; its tail is printed on failure."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    logdir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(logdir, exist_ok=True)
    procs, logs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        logs.append(open(os.path.join(logdir, f"bench_rank{r}.err"), "w+"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=logs[-1], text=True))
    import threading
    out = []
    reader = threading.Thread(target=lambda: out.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed_at = None
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        if failed_at is None and any(c not in (None, 0) for c in codes):
            failed_at = time.monotonic()
        if failed_at is not None and time.monotonic() - failed_at > grace_s:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            time.sleep(2.0)
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(poll_s)
    reader.join(timeout=5)
    rc = max(abs(p.returncode) for p in procs)
    sys.stdout.write("".join(out))
    sys.stdout.flush()
    if rc:
        for r, f in enumerate(logs):
            f.seek(0)
            tail = f.read()[-1500:]
            sys.stderr.write(f"[bench launcher] rank {r} exit code {procs[r].returncode}; stderr tail:\n{tail}\n")
    for f in logs:
        f.close()
    return rc


# ------------------------------------------------------------------------------------------------ roofline helpers
def gemm_algorithmic_flops(n=N_TOTAL, dim=1024, depth=22, ff_mult=2, mel=100, nfe=STEPS_NFE, branches=2, unett=False):
    """FLOPs the GEMM kernel class must do per unit inside the ODE loop (SURVEY section 8(d) per-token MACs, real
    dims, no padding, the split-bf16 x3 NOT counted, step-invariant work hoisted out NOT counted).  unett: E2-TTS (ff_mult 4, one
    Linear(2 dim -> dim) U-skip projection in the second half of the layers, one more row per sequence: the time token)."""
    per_tok = depth * (4 * dim * dim + 2 * ff_mult * dim * dim)      # qkv + out + ff1 + ff2 = 184.55 M at F5-Base
    if unett:
        per_tok += (depth // 2) * 2 * dim * dim
        n = n + 1
    per_tok += mel * dim                                              # x part of the input projection
    per_tok += 2 * (dim // 16) * 31 * dim                             # conv_pos_embed, 2 grouped convs
    per_tok += dim * mel                                              # proj_out
    return 2.0 * per_tok * n * branches * nfe


def attn_algorithmic_flops(n=N_TOTAL, dim=1024, depth=22, nfe=STEPS_NFE, branches=2, unett=False):
    if unett:
        n = n + 1
    return 2.0 * depth * 2 * n * dim * n * branches * nfe


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sd, vsd, cond, text, y0, n_threads, full=False):
    """The oracle (a port of the reference's fp32 CPU path) on the host cores, bounded sample: 4 of the 32 Euler
    steps with CFG (8 DiT forwards at N = 1404) + one Vocos decode; the ODE part is scaled x8 to the 32-step job.  full (--cpu-full):
    all 32 steps, nothing scaled -- the confirmation run SURVEY 8(d) asks for once."""
    import torch
    from oracle import dit_oracle as O
    from oracle import vocos_oracle as V
    torch.set_num_threads(n_threads)
    sample_steps = STEPS_NFE if full else 4
    t0 = time.time()
    out, _ = O.cfm_sample(sd, O.F5_BASE, cond, text, N_TOTAL, steps=sample_steps, cfg_strength=CFG, sway_sampling_coef=SWAY, y0=y0,
                          keep_trajectory=False)
    t_ode = time.time() - t0
    t0 = time.time()
    V.vocos_decode(vsd, out[:, N_REF:].permute(0, 2, 1))
    t_voc = time.time() - t0
    wall = t_ode * (STEPS_NFE / sample_steps) + t_voc
    return {"value": round((N_TOTAL - N_REF) / wall, 3), "unit": "mel-frames/s", "cores": n_threads, "cpu_model": cpu_model_name(), "kind": "port",
            "sample": (f"all 32 Euler steps with CFG (64 DiT forwards, N=1404) = {t_ode:.1f} s, + 1 Vocos decode = {t_voc:.2f} s: the whole job, nothing scaled" if full else
                       f"{sample_steps} of 32 Euler steps with CFG ({2 * sample_steps} DiT forwards, N=1404) = {t_ode:.1f} s scaled x{STEPS_NFE // sample_steps}, + 1 Vocos decode = {t_voc:.2f} s"),
            "rtf": round(wall / ((N_TOTAL - N_REF - 1) * 256 / 24000.0), 3)}


def pmc_traffic():
    """Memory-side bytes per launch of the dominant GEMM kernel from the committed rocprofv3 --pmc passes (not collectable live: the
    counters need the profiler).  None when the file is absent or was taken for another kernel build."""
    for path in TRAFFIC_FILES:
        try:
            d = json.load(open(path))
            return float(d["bytes_per_launch"]), d.get("source", path)
        except (OSError, ValueError, KeyError):
            continue
    return None, None


# ------------------------------------------------------------------------------------------------ one rank
def run_rank(args):
    import torch
    import torch.distributed as dist

    if os.environ.get("F5HIP_BENCH_WATCHDOG"):          # dump every thread's stack and exit if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["F5HIP_BENCH_WATCHDOG"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # one process per GPU; F5HIP_DIST_BACKEND=gloo + several ranks on one card is only for rehearsing the N > 1 path on a 1-GPU box
    backend = os.environ.get("F5HIP_DIST_BACKEND", "nccl")
    fake = os.environ.get("F5HIP_BENCH_FAKE") == "1"    # CPU test of the launcher / process group only: no GPU, no kernels
    if fake:
        backend = "gloo"
        if os.environ.get("F5HIP_BENCH_FAKE_FAIL_RANK") == str(rank):   # launcher test: this rank dies before its first collective
            sys.stderr.write("fake failure of rank %d\n" % rank)
            return 3
    dev_index = 0 if fake else local_rank % max(1, torch.cuda.device_count())
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{dev_index}"))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    n_gpus = dist.get_world_size() if dist.is_initialized() else 1
    if fake:
        t = torch.tensor([float(rank + 1)])
        if world > 1:
            dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"metric": "launcher self-test (no GPU work)", "value": float(t), "n_gpus": n_gpus, "steps": args.steps,
                              "warmup": args.warmup, "mode": args.mode}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return 0
    torch.cuda.set_device(dev_index)
    dev = torch.device(f"cuda:{dev_index}")

    from tts_indic_server_f5_amd import _lib, synth
    from tts_indic_server_f5_amd.model import E2TTS_BASE, F5TTS_BASE, F5HipModel
    from tts_indic_server_f5_amd.sharding import broadcast_ref_latents, gather_waves, shard_units
    from tts_indic_server_f5_amd.vocoder import F5HipVocos

    e2 = args.arch == "e2"
    nfe = args.nfe if args.nfe is not None else STEPS_NFE
    unit_frames = E2_FRAMES if e2 else N_TOTAL
    n_gen_ids = N_GEN_IDS * (unit_frames - N_REF) // (N_TOTAL - N_REF)            # text in proportion to the generated audio
    sd, vsd = (synth.unett_state_dict() if e2 else synth.dit_state_dict()), synth.vocos_state_dict()
    model = F5HipModel(E2TTS_BASE if e2 else F5TTS_BASE, sd, gemm_planes=args.gemm_planes, device=dev)
    vocos = F5HipVocos(vsd, gemm_planes=min(args.gemm_planes, 2), device=dev)
    bigv = None
    if args.vocoder == "bigvgan":
        from tts_indic_server_f5_amd.vocoder import F5HipBigVGAN
        bigv = F5HipBigVGAN(synth.bigvgan_state_dict(), gemm_planes=args.vocoder_planes, device=dev)

    # ---- the units of one step: (global index, total frames).  Every rank derives the same list and takes its share.
    n_units = args.total_batch if args.mode == "strong" else args.batch * world
    if args.ragged:
        g = torch.Generator().manual_seed(99)
        gen_frames = (torch.rand(n_units, generator=g) * (14.0 - 6.0) + 6.0) * 24000.0 / 256.0
        frames = [N_REF + int(f) for f in gen_frames]
    else:
        frames = [unit_frames] * n_units
    if args.mode == "strong":
        mine = shard_units(frames, world)[rank]
    else:
        mine = list(range(rank * args.batch, (rank + 1) * args.batch))
    B = len(mine)
    my_frames = [frames[u] for u in mine]

    # rank 0 owns the reference-audio latents; every unit has its own gen text + noise (seeded by its global index)
    gc = torch.Generator().manual_seed(14)
    cond0 = torch.randn(N_REF + 1, 100, generator=gc).to(dev) if rank == 0 else None
    ref_ids0 = synth.text_ids(N_REF_IDS, 0)[0].to(dev) if rank == 0 else None
    gen_ids = [synth.text_ids(N_REF_IDS, n_gen_ids, seed=synth.SEED_TEXT + u)[0][N_REF_IDS:].to(dev) for u in mine]
    y0 = [synth.noise(frames[u], u).to(dev) for u in mine]                       # device-resident before the timed region

    def one_step(exchange=True):
        # exchange=False (rank 0's untimed profiling pass) must not enter a collective the other ranks never join
        cond, ref_ids = broadcast_ref_latents(cond0, ref_ids0, dev) if exchange else (cond0, ref_ids0)
        waves = []
        if B:
            text = torch.stack([torch.cat([ref_ids, g]) for g in gen_ids])
            out, _ = model.sample(cond[None].expand(B, -1, -1), text, torch.tensor(my_frames), steps=nfe, cfg_strength=CFG,
                                  sway_sampling_coef=SWAY, y0=y0)
            if len(set(my_frames)) == 1:
                mel = out[:, N_REF:, :].permute(0, 2, 1)
                w = bigv(mel) if bigv is not None else vocos.decode(mel)
                waves = list(w.reshape(B, -1))
            else:
                for i, n in enumerate(my_frames):
                    mel = out[i:i + 1, N_REF:n, :].permute(0, 2, 1)
                    waves.append((bigv(mel) if bigv is not None else vocos.decode(mel)).reshape(-1))
        if args.mode == "strong" and exchange and world > 1:
            flat = torch.cat(waves) if waves else torch.zeros(0, device=dev)
            got = gather_waves(flat)                                             # variable-length gather to rank 0 over RCCL
            return [w.cpu() for w in got] if got is not None else []
        return [w.cpu() for w in waves]

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
        waves = one_step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    gen_total = sum(f - N_REF for f in frames)                                   # generated mel frames of ALL ranks per step
    value = gen_total * args.steps / dt
    audio_s = sum((f - N_REF - 1) * 256 for f in frames) / 24000.0

    if rank == 0:
        # per-kernel-class durations: one more pass with HIP events around every launch on the launch stream
        import ctypes as C
        L = _lib.lib()
        L.f5hip_set_profiling(1)
        one_step(exchange=False)
        torch.cuda.synchronize()
        prof = {}
        for cls in ("gemm", "attn", "ln", "other", "vocos"):
            ms, n = C.c_double(0), C.c_int64(0)
            L.f5hip_get_profile(cls.encode(), C.byref(ms), C.byref(n))
            prof[cls] = {"total_ms": round(ms.value, 3), "launches": n.value}
        L.f5hip_set_profiling(0)
        # What the event pairs cost: the GPU is >= 98 % busy in the timed (uninstrumented) steps (rocprofv3: ~0.5 us between launches), so
        # the excess of the instrumented pass's spans over one timed step is the spans' own cost; per span = excess / spans.  (64 EMPTY spans
        # back to back measure 4.5 us each, twice what a span costs between kernels -- 2.1-2.3 us -- so that is not used.)  The per-launch
        # durations net of it agree with the rocprofv3 kernel trace of the same command within ~5 % (profiles/).
        n_spans = sum(v["launches"] for v in prof.values())
        span_us = max(0.0, (sum(v["total_ms"] for v in prof.values()) - dt / args.steps * 1e3) / max(1, n_spans)) * 1e3
        for cls in prof:
            prof[cls]["net_ms"] = round(max(0.0, prof[cls]["total_ms"] - prof[cls]["launches"] * span_us * 1e-3), 3)
        g = prof["gemm"]
        # the profiled pass also ran the hoisted / Vocos GEMMs; their share of launches and time is < 2 %
        arch_kw = dict(depth=24, ff_mult=4, unett=True) if e2 else {}
        gemm_flops = sum(gemm_algorithmic_flops(n=f, nfe=nfe, **arch_kw) for f in my_frames)   # rank 0's units
        achieved_raw = gemm_flops / (g["total_ms"] * 1e-3) / 1e12 if g["total_ms"] > 0 else 0.0
        achieved = gemm_flops / (g["net_ms"] * 1e-3) / 1e12 if g["net_ms"] > 0 else 0.0
        att = prof["attn"]
        attn_fl = sum(attn_algorithmic_flops(n=f, nfe=nfe, **({"depth": 24, "unett": True} if e2 else {})) for f in my_frames)
        attn_tf = attn_fl / (att["net_ms"] * 1e-3) / 1e12 if att["net_ms"] > 0 else 0.0
        traffic, traffic_src = pmc_traffic() if (args.gemm_planes == 3 and B == 1 and not args.ragged and not e2) else (None, None)
        coll = "RCCL" if backend == "nccl" else backend     # (gloo only when rehearsing the N > 1 path on a box with fewer GPUs than ranks)
        par = f"utterance-sharded x{n_gpus}, {coll} broadcast of ref latents" + (f", LPT dealing + {coll} gather of the waveforms to rank 0" if args.mode == "strong" else "")
        result = {
            "metric": "mel-frames/sec + RTF, F5-TTS-Base 32-NFE, 10s utterance" if not e2 else f"mel-frames/sec + RTF, E2-TTS-Base {nfe}-NFE, 20 s chunks (BASELINE configs[4])",
            "value": round(value, 1),
            "unit": "mel-frames/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.mode,
            "vs_baseline": None, "dtype": "fp16/bf16 MFMA, fp32 accumulate" if args.gemm_planes == 3 else "bf16", "data": "synthetic",
            "rtf": round(dt / args.steps / audio_s, 6),
            "config": {"workload": f"{'E2-TTS-Base (UNetT)' if e2 else 'F5-TTS-Base'}, {nfe} NFE + CFG=2.0 + sway -1, {('BigVGAN (' + {1: 'bf16', 2: 'split bf16', 3: 'fp16 fast mode'}[args.vocoder_planes] + ' convs)') if bigv is not None else 'Vocos'}, {n_units} x "
                                   f"{'U(6 s, 14 s)' if args.ragged else ('20 s' if e2 else '10 s')} {'chunk' if e2 else 'utterance'}{'s' if n_units > 1 else ''} per step over {n_gpus} GPU{'s' if n_gpus > 1 else ''}"
                                   f" ({B} on rank 0; N={unit_frames} = 468 reference + {unit_frames - N_REF} generated frames)",
                       "gemm_mode": GEMM_MODES[args.gemm_planes],
                       "attention": "fp16 MFMA operands (q, k, v, p), fp32 scores / softmax / accumulation", "parallelism": par},
            "roofline": {"bound": "mfma", "kernel": "gemm5_kernel / gemm6_kernel (fp16 transformer-block GEMMs: exact-fit tiles at one utterance, 256 x 256 ping-pong tiles in batch mode) + gemm_kernel / gemm3_kernel (bf16x3 state GEMMs): all GEMM launches of rank 0",
                         "timing": "HIP events around every launch in an extra instrumented pass of rank 0, on the launch stream; the cost of an event pair (event_span_us = the excess of that pass over one timed step, per span) "
                                   "is subtracted per launch (kernel_ms.*.net_ms); frac_raw_events is the figure without that correction; the rocprofv3 kernel-trace summary of the same command is under profiles/",
                         "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "frac_raw_events": round(achieved_raw / PEAK_BF16_TFLOPS, 4), "event_span_us": round(span_us, 3),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "avg_launch_us": round(g["net_ms"] * 1e3 / max(1, g["launches"]), 2), "launches": g["launches"],
                         "executed_mfma_x": {1: 1, 2: 3, 3: 1.02}[args.gemm_planes],
                         "attn_tflops": round(attn_tf, 1), "attn_frac": round(attn_tf / PEAK_BF16_TFLOPS, 4)},
            "kernel_ms": prof,
        }
        if bigv is not None and not args.ragged and prof["vocos"]["total_ms"] > 0:
            # north_star: "achieved HBM GB/s on vocoder conv".  Algorithmic bytes = the ideal fused fp32 activation traffic of one 936-frame
            # BigVGAN v2 decode (9.1 GB) x the utterances rank 0 decoded, over the decode's HIP-event time.
            v_gbs = 9.1 * B / (prof["vocos"]["total_ms"] * 1e-3)
            result["vocoder_roofline"] = {"bound": "hbm", "kernel": "BigVGAN v2 decode: conv5 / implicit-GEMM convolutions + aa_snake2 + block mean + conv_post",
                                          "achieved": round(v_gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(v_gbs / 8000.0, 4),
                                          "algorithmic_bytes_per_utterance": 9.1e9, "utterances": B, "decode_ms": prof["vocos"]["total_ms"],
                                          "conv_precision": {1: "bf16", 2: "split bf16 (parity mode)", 3: "fp16 (fast mode, outside the 1e-4 waveform bound)"}[args.vocoder_planes]}
        if not args.no_cpu_baseline and world == 1 and B == 1 and bigv is None and not args.ragged and not e2 and nfe == STEPS_NFE:
            n_threads = min(len(os.sched_getaffinity(0)), 32)
            result["cpu_baseline"] = cpu_baseline(sd, vsd, cond0.cpu()[None], torch.cat([ref_ids0, gen_ids[0]]).cpu()[None], y0[0].cpu()[None], n_threads, full=args.cpu_full)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    sys.exit(run_rank(args))


if __name__ == "__main__":
    main()
