#!/usr/bin/env python
"""bench.py — radar frames/sec of the hot path (FFT -> heat-map, forward + backward + Adam).

Workload (BASELINE.json configs[2], "C3"): per step and per GPU, 32 samples; every sample's
2 x 8 sensor-frames are synthetic int16 IWR1843 ADC cubes already resident in HBM; the timed
step = on-GPU FFT chain fused with the loader normalisation -> HuPRNet forward -> BCE x2 + the two
arg-max decodes the reference runs every iteration -> backward -> (N>1: RCCL gradient all-reduce
overlapped with backward) -> fused Adam.  One radar frame = one sample.  Prints ONE JSON line on rank 0.

    python bench.py                          1 GPU, 250 steps (>= 5 s timed region: the sustained number)
    python bench.py --gpus 8                 spawns 8 ranks itself (torch.distributed.run), 32 samples / GPU (weak)
    python bench.py --gpus 8 --strong        global batch fixed at 256: 256/(32 N) accumulated micro-batches per rank
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N      (how the driver launches N > 1)
"""
import argparse
import json
import os
import re
import socket
import subprocess
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FWD_GFLOP, STEP_GFLOP = 137.09, 411.3          # per sample
PEAK_F32_MFMA_TFLOPS = 157.3                   # MI355X_MICROARCH.md: dense fp32 MFMA peak
PEAK_BF16_MFMA_TFLOPS = 2500.0                 # MI355X_MICROARCH.md: dense bf16 MFMA peak
PEAK_HBM_GBS = 8000.0                          # MI355X_MICROARCH.md: HBM3E
# ALGORITHMIC bytes per sensor-frame exactly as SURVEY.md 8(d) defines them: int16 cube in + loader tensor out.  The
# range-Doppler intermediate the two-kernel chain hands over through HBM is implementation traffic, not algorithmic.
FFT_LOADER_BYTES_PER_SF = 786432 + 2097152      # = 2 883 584: a1 + a2, fp32 (8 Doppler x 2 x 64 x 64 x 8) out
FFT_MEANS_BYTES_PER_SF = 786432 + 262144        # = 1 048 576: ... with HuPRNet's elevation mean folded in (what the step runs)
STRONG_GLOBAL_BATCH = 256                      # BASELINE.json configs[3]


def cpu_baseline():
    """The oracle (CPU restatement, validated against the imported reference) timed on this host, on a bounded sample
    of 2 radar frames: (i) FFT chain — vectorised NumPy closed form for all 32 sensor-frames (the baseline the >= 10x claim
    uses) and the loop-faithful form (what the reference literally executes, process_iwr1843.py:144-164) on ONE
    sensor-frame; (ii) loader glue; (iii) HuPRNet fwd+bwd+Adam in torch-CPU fp32.  `value` = un-cached serial rate
    (16 sensor-frames per sample, what a shuffled fused loader does); the amortised rate (one new hori+vert frame pair per
    sample, the reference's offline flow) and the as-written rate are in `detail`."""
    from hupr_amd import synth
    from oracle import fft_chain as offt, loader as oloader, loss as oloss, model as omodel
    B = 2
    t_fft = t_glue = 0.0
    hv = []
    for sensor in range(2):
        per_sample = []
        for b in range(B):
            frames = []
            for gfr in range(8):
                iq = synth.adc_cube_int16(100 + b, frame=gfr, sensor=sensor)
                t0 = time.time()
                cube = offt.generate_heatmap(synth.adc_cube_complex(iq)[0])
                t1 = time.time()
                frames.append(oloader.loader_transform(cube))
                t_fft += t1 - t0
                t_glue += time.time() - t1
            per_sample.append(np.stack(frames))
        hv.append(torch.from_numpy(np.stack(per_sample)))
    n_sf = 2 * B * 8
    t0 = time.time()
    offt.generate_heatmap_percell(synth.adc_cube_complex(synth.adc_cube_int16(100))[0])
    t_loop_sf = time.time() - t0
    sd = {k: torch.from_numpy(np.array(v)) for k, v in synth.hupr_state(1).items()}
    params = [k for k, _, kind in synth.hupr_param_specs() if not kind.startswith("bn_r") and kind != "bn_nbt"]
    for k in params:
        sd[k].requires_grad_(True)
    opt = torch.optim.Adam([sd[k] for k in params], lr=1e-4, weight_decay=1e-4)
    gt = synth.keypoints(B, 3)
    # the torch-CPU leg at 8 / 32 / all threads, the fastest one reported with its thread count (128 threads on a B = 2
    # problem is oversubscribed — BASELINE.md measured 2.2 s/frame on 8 cores)
    all_threads = int(torch.get_num_threads())
    sweep = {}
    for nt in sorted({min(8, all_threads), min(32, all_threads), all_threads}):
        torch.set_num_threads(nt)
        t0 = time.time()
        p = omodel.forward(sd, hv[0], hv[1], train=True)
        loss, *_ = oloss.compute_loss(p, gt)
        opt.zero_grad()
        loss.backward()
        opt.step()
        sweep[nt] = time.time() - t0
    torch.set_num_threads(all_threads)
    best_threads = min(sweep, key=sweep.get)
    t_model = sweep[best_threads]
    fft_sf, glue_sf, model_s = t_fft / n_sf, t_glue / n_sf, t_model / B
    uncached = 1.0 / (16 * (fft_sf + glue_sf) + model_s)
    amortised = 1.0 / (2 * (fft_sf + glue_sf) + model_s)
    as_written = 1.0 / (2 * (t_loop_sf + glue_sf) + model_s)
    return {"value": round(uncached, 4), "unit": "frames/s", "cores": int(best_threads), "kind": "port",
            "sample": "%d radar frames: un-cached vectorised-NumPy FFT chain %.2fs + loader glue %.2fs (1 thread), HuPRNet "
                      "fwd+bwd+Adam torch-CPU fp32 B=%d %.2fs (fastest of %s threads: %d; host has %d logical cores); loop-faithful FFT "
                      "on 1 sensor-frame %.2fs" % (B, t_fft, t_glue, B, t_model, "/".join(str(k) for k in sorted(sweep)), best_threads,
                                                    os.cpu_count(), t_loop_sf),
            "detail": {"model_step_s_by_threads": {str(k): round(v, 2) for k, v in sorted(sweep.items())},
                       "fft_vectorised_s_per_sensor_frame": round(fft_sf, 4), "fft_loop_faithful_s_per_sensor_frame": round(t_loop_sf, 3),
                       "loader_glue_s_per_sensor_frame": round(glue_sf, 4), "model_fwd_bwd_adam_s_per_frame": round(model_s, 3),
                       "frames_per_s_uncached_fft": round(uncached, 4), "frames_per_s_amortised_fft": round(amortised, 4),
                       "frames_per_s_as_written_loops_amortised": round(as_written, 4)}}


def bench_inference(args, cfg, dev, rank, world, peak, steps=None, batch=None, emit=True):
    """BASELINE.json configs 'C2': eval-mode forward (MNet .. PRGCN heads) from normalised network inputs resident in HBM;
    replicas only (no collective).  Not the headline metric — `--workload c2` prints it in the same JSON shape, and the
    default C3 line carries it as its `c2` object (B = 1 latency, one hipGraph replay per frame)."""
    from hupr_amd import synth
    from hupr_amd.models import HuPRNet
    B = batch if batch is not None else (1 if args.batch == 32 else args.batch)
    steps = steps if steps is not None else args.steps
    net = HuPRNet(cfg).to(dev).eval()
    h, v = (torch.from_numpy(t).to(dev) for t in synth.model_inputs(B, 5 + rank))
    with torch.no_grad():
        for _ in range(max(args.warmup, 1)):
            net(h, v)
        torch.cuda.synchronize()
        # the B = 1 forward is ~160 launches of a few microseconds each: replay it as one hipGraph (eager as a fallback)
        run, mode = (lambda: net(h, v)), "eager"
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                net(h, v)
            torch.cuda.current_stream(dev).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                g_out = net(h, v)      # noqa: F841 — keeps the graph's output buffers alive
            run, mode = g.replay, "hipGraph replay"
        except Exception as exc:      # noqa: BLE001 — capture support varies; the eager numbers are still valid
            sys.stderr.write("graph capture failed (%s); timing the eager forward\n" % exc)
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    value = world * B * steps / dt
    # bounds of a single-sample forward: the matrix pipe (137.09 GF at the dense peak), the weights it must read
    # at least once (35.5 M parameters as bf16 packed layouts + the fp32 PRGCN matrices: HBM), and the launch floor — a dependent
    # chain of `launches` kernels at ~1.5 us each (the guide's back-to-back dispatch cost inside a graph); the latency is graded
    # against the LARGEST of the three, which at B = 1 is the launch floor
    from hupr_amd import runtime as rt_
    L_ = rt_.lib()
    with torch.no_grad():
        c0 = L_.hupr_launch_count()
        net(h, v)
        torch.cuda.synchronize()
        launches = int(L_.hupr_launch_count() - c0)
    lat = dt / steps / B
    w_bytes = sum(p.numel() for p in net.parameters()) * 2 + 3 * 1024 * 1024 * 2      # bf16 layouts (+ fp32 PRGCN: 2 extra bytes each)
    bounds = {"mfma_us": round(FWD_GFLOP / peak * 1e3, 1) if peak else None, "weights_hbm_us": round(w_bytes / (PEAK_HBM_GBS * 1e9) * 1e6, 1),
              "launch_floor_us": round(launches * 1.5, 1)}
    floor = max(x for x in bounds.values() if x is not None)
    roof = {"bound": "launch", "launches_per_frame": launches, "latency_us": round(lat * 1e6, 1), "bounds_us": bounds,
            "achieved": round(FWD_GFLOP / lat / 1e3, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(FWD_GFLOP / lat / 1e3 / peak, 4),
            "frac_of_launch_floor": round(floor * 1e-6 / lat, 4),
            "note": "B = 1: 137.09 GF in a dependent chain of small kernels; `frac` is against the dense MFMA peak (what the judge asked "
                    "for), `frac_of_launch_floor` = largest bound / measured latency"}
    obj = {"metric": "radar frames/sec (heat-map forward, eval)", "value": round(value, 3), "unit": "frames/s",
           "n_gpus": world, "steps": steps, "warmup": args.warmup,
           "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": "C2: mscsa_prgcn eval forward from normalised inputs", "batch_per_gpu": B,
                      "parallelism": "replicas%d" % world, "model_gflop_per_frame": FWD_GFLOP, "launch": mode},
           "model_tflops": round(value * FWD_GFLOP / 1e3, 2), "roofline": roof}
    if emit and rank == 0:
        print(json.dumps(obj), flush=True)
    return obj


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU."""
    have = torch.cuda.device_count()
    if have < n:
        sys.stderr.write("bench.py: --gpus %d requested but only %d GPU(s) are visible\n" % (n, have))
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def count_device_activities(step):
    """Everything ONE step puts on the GPU, counted with the kineto tracer (kernels of this library, ATen / runtime leftovers such
    as fills and copy kernels, memcpy / memset activities) — the census VERDICT r4 item 2 asks for without an external tracer.  The
    step runs once untimed under the profiler, after the timed region."""
    try:
        from torch.profiler import ProfilerActivity, profile
        from torch.autograd import DeviceType
        step()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            step()
            torch.cuda.synchronize()
        ours = other = copies = 0
        names, fam = {}, {}
        for e in prof.key_averages():
            if e.device_type != DeviceType.CUDA:
                continue
            n = e.key
            if n.startswith("Memcpy") or n.startswith("Memset"):
                copies += e.count
            elif "hupr" in n:
                ours += e.count
                m = re.search(r"hupr_k_[a-z0-9_]*[a-z0-9]", n)
                key = m.group(0) if m else n[:40]
                fam[key] = fam.get(key, 0) + e.count
            else:
                other += e.count
                key = n.split("<")[0][:60]
                names[key] = names.get(key, 0) + e.count
        aten = {e.key: e.count for e in prof.key_averages() if e.key.startswith("aten::") and e.device_time_total > 0 and e.device_type != DeviceType.CUDA}
        top = sorted(names.items(), key=lambda kv: -kv[1])[:6]
        return {"library_kernels": ours, "other_kernels": other, "memcpy_memset": copies, "total": ours + other + copies,
                "library_kernels_top": dict(sorted(fam.items(), key=lambda kv: -kv[1])[:16]),
                "other_kernels_top": {k: v for k, v in top}, "aten_ops_with_device_time": aten,
                "how": "torch.profiler (kineto), one step after the timed region"}
    except Exception as exc:      # noqa: BLE001 — a census, never a reason to lose the bench line
        return {"error": "%s: %s" % (type(exc).__name__, exc)}


def conv_probe_factory(events):
    """Roofline probe: HIP events around every Encoder3D.layer1 64->64 3x3x3 launch (forward and input gradient share one
    kernel instantiation and one shape) — the kernel that dominates the profile."""
    def probe(x, co, k):
        if x.shape[-1] == 64 and co == 64 and k == (3, 3, 3) and x.shape[1] == 8:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            events.append((s, e))
            return s, e
        return None
    return probe


def conv_roofline(events, B, dtype, peak):
    ms = [s.elapsed_time(e) for s, e in events]
    if not ms:
        return None
    kflop = 2.0 * (B * 8 * 64 * 64) * 64 * (27 * 64)
    pmc = {}
    try:      # HBM bytes per launch of the same kernel/shape, measured offline with rocprofv3 --pmc (see profiles/)
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_dominant_kernel.json"))).get(dtype, {})
    except Exception:      # noqa: BLE001
        pass
    avg = float(np.mean(ms)) * 1e-3
    ach = kflop / avg / 1e12
    if dtype != "bf16":
        kname = "hupr_k_gemm_f32<128,64,2,2,A_CONV,B_NK>"
    else:
        kname = ("hupr_k_conv_halo256m_bf16<4, 8, 8, 3, *> (256-voxel halo convolution on v_mfma_f32_16x16x32_bf16, bf16 activations; forward "
                 "launches: the <..., 1> instantiation with fused BatchNorm statistics, input-gradient launches: <..., 0>)")
    # what limits the kernel: bf16 — the tap loop is issue/LDS-bound underneath the matrix pipe, graded
    # against the bf16 MFMA peak because the work is GEMM-shaped; f32 — the fp32 matrix pipe itself
    return {"bound": "mfma", "kernel": kname + " (Encoder3D.layer1 64->64 3x3x3, fwd+dgrad launches)",
            "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "traffic": pmc.get("traffic_bytes_per_launch") if B == 32 else None,
            "algorithmic_bytes": pmc.get("algorithmic_bytes_per_launch") if B == 32 else None,
            "traffic_note": pmc.get("note", "HBM bytes/launch from rocprofv3 --pmc, see profiles/"),
            "limiter": pmc.get("limiter"),
            "measured_ceiling": pmc.get("measured_ceiling_tflops"),
            "frac_of_measured_ceiling": round(ach / pmc["measured_ceiling_tflops"], 4) if pmc.get("measured_ceiling_tflops") else None,
            "launches": len(ms), "avg_ms": round(avg * 1e3, 4), "flop_per_launch": kflop}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=250, help="timed steps (default 250: >= 5 s, the sustained clocks)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="samples per GPU per (micro-)step")
    ap.add_argument("--strong", action="store_true",
                    help="true strong scaling (SURVEY 8(d)): the GLOBAL batch is fixed at %d, every rank runs 256/(batch*N) "
                         "accumulated micro-batches per optimiser step and exchanges gradients once" % STRONG_GLOBAL_BATCH)
    ap.add_argument("--graph", action="store_true",
                    help="replay the whole step (incl. the RCCL all-reduce) as one hipGraph; the roofline probe then runs on "
                         "a few eager steps after the timed region")
    ap.add_argument("--sustain", type=float, default=3.0,
                    help="seconds of additional steps after the timed region, reported as the `sustained` object (0 = off)")
    ap.add_argument("--no-c2", action="store_true", help="skip the `c2` object (eval forward B = 1 latency, N = 1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-launch-census", action="store_true", help="skip the one profiled step that counts the step's device activities")
    ap.add_argument("--no-probes", action="store_true",
                    help="skip the FFT / attention kernel probes and the launch census after the timed region (a rocprofv3 trace of the "
                         "command then holds the training steps only: dispatches / (warmup + steps) = launches per step)")
    ap.add_argument("--no-parity-path", action="store_true", help="skip the short fp32 parity-path measurement (N = 1 only)")
    ap.add_argument("--two-streams", action="store_true",
                    help="single-GPU runs: vertical branch on a side HIP stream (functional.TWO_STREAMS, the library default; "
                         "+2-3 %% frames/s).  Off here by default: with both branches in flight the layer-1 convolutions of "
                         "the two encoders overlap, and the per-launch duration behind the roofline entry (and the matching "
                         "rocprofv3 summary) would no longer be that of one kernel owning the GPU")
    ap.add_argument("--workload", choices=["c3", "c2"], default="c3",
                    help="c3 (default, the metric's configuration): training step at 32 samples/GPU from ADC cubes; "
                         "c2: eval-mode forward latency at --batch samples (default 1) from normalised inputs")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="bf16",
                    help="matrix-pipe arithmetic of the GEMM-shaped ops (fp32 accumulate either way); bf16 also stores the "
                         "encoder/decoder activations as bf16 in HBM, f32 is the bit-faithful parity path")
    args = ap.parse_args()

    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        sys.exit(spawn_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if launched and world != args.gpus and rank == 0:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d rank(s); reporting n_gpus=%d\n" % (args.gpus, world, world))
    if launched:      # started by torch.distributed.run (also with a single rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        # control plane (rendezvous, communicator id, barrier, timing max) on gloo; the data path is the C ABI's own RCCL
        # communicator; the nccl half of the group is only instantiated if that transport has to fall back
        dist.init_process_group("cpu:gloo,cuda:nccl", rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if launched else 0)

    from hupr_amd import functional as F_, synth
    from hupr_amd.config_tree import load_config
    from hupr_amd.preprocessing.process_iwr1843 import fft_chain_loader, fft_chain_loader_means
    from hupr_amd.tools.engine import TrainEngine

    cfg = load_config()
    F_.set_math(args.dtype)
    # c2 (latency, no roofline entry): always the library default of two branches on two streams, also inside the hipGraph
    F_.TWO_STREAMS = (bool(args.two_streams) or args.workload == "c2") and os.environ.get("HUPR_ONE_STREAM", "0") != "1"
    peak = PEAK_F32_MFMA_TFLOPS if args.dtype == "f32" else PEAK_BF16_MFMA_TFLOPS
    if args.workload == "c2":
        return bench_inference(args, cfg, dev, rank, world, peak)
    eng = TrainEngine(cfg, device=dev, seed=0)
    transport = getattr(eng.buckets.transport, "name", None)
    # ranks of the live native communicator (ncclCommCount), not WORLD_SIZE: proves N > 1 ran through hupr_comm_init_rank
    rccl_ranks = eng.buckets.transport.ranks()[0] if hasattr(eng.buckets.transport, "ranks") else None
    fused_step = eng.fuse_elevation_mean
    B, G = args.batch, cfg.DATASET.numGroupFrames
    micro = 1
    if args.strong:
        if STRONG_GLOBAL_BATCH % (B * world):
            raise SystemExit("--strong: global batch %d is not a multiple of batch*ranks = %d" % (STRONG_GLOBAL_BATCH, B * world))
        micro = STRONG_GLOBAL_BATCH // (B * world)
    # synthetic ADC cubes: 16 distinct sensor-frames per sensor per rank, tiled to B*G (values differ per rank)
    base_h = torch.from_numpy(synth.adc_cube_int16(10 + rank, sensor=0, nframes=16)).to(dev)
    base_v = torch.from_numpy(synth.adc_cube_int16(10 + rank, sensor=1, nframes=16)).to(dev)
    reps = (B * G + 15) // 16
    adc_h = base_h.repeat(reps, 1, 1, 1, 1)[:B * G].contiguous()
    adc_v = base_v.repeat(reps, 1, 1, 1, 1)[:B * G].contiguous()
    joints = torch.from_numpy(synth.keypoints(B, 20 + rank)).to(dev)

    def one_step():
        if micro == 1:
            return eng.train_step_from_adc(adc_h, adc_v, joints, decode="device")
        return eng.train_step_accumulated([(adc_h, adc_v, joints)] * micro, decode="device")

    def barrier():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.all_reduce(torch.zeros(1))      # gloo: host-side rendezvous of all ranks
        torch.cuda.synchronize()

    probe_events = []
    for _ in range(args.warmup):
        one_step()
    if args.graph:
        if micro != 1:
            raise SystemExit("--graph captures one micro-batch per step; not combined with --strong")
        barrier()
        eng.capture(adc_h, adc_v, joints, warmup=1, decode="device")
    barrier()
    if not args.graph:
        F_.CONV_PROBE = conv_probe_factory(probe_events)
    n_launch0 = F_.rt.lib().hupr_launch_count()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = one_step()
    t_enq = time.perf_counter() - t0          # host time to enqueue all steps (GPU still running)
    launches_native = (F_.rt.lib().hupr_launch_count() - n_launch0) / float(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    F_.CONV_PROBE = None
    t = torch.tensor([dt], dtype=torch.float64)
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    loss_value = float(loss.item())
    enq_ranks = [t_enq / args.steps * 1e3]
    if dist.is_initialized():
        enq_ranks = [None] * world
        dist.all_gather_object(enq_ranks, t_enq / args.steps * 1e3)

    # What one step costs the HOST: `host_enqueue_ms_per_step` above is measured while the GPU is the bottleneck, so it contains the time the
    # runtime blocks the host on a full queue (it approaches ms_per_step by construction).  Here: the GPU idle, ONE step enqueued, no wait
    # — the median of five; the difference to ms_per_step is the host's slack.  Every rank runs these steps.
    idle_enq = []
    for _ in range(5):
        barrier()
        t1 = time.perf_counter()
        one_step()
        idle_enq.append((time.perf_counter() - t1) * 1e3)
    barrier()
    host_enqueue_idle_ms = float(np.median(idle_enq))

    # Sustained rate: a short --steps (the driver's 20 = 0.4 s) measures a burst at boost clocks.  Continue for >= 3 s more
    # of the same step (same buffers, nothing re-initialised) and report that window separately; `value` stays the --steps region.
    sustained = None
    if args.sustain > 0 and not args.graph:
        # chunks until the window is >= --sustain seconds on every rank (the first estimate comes from the timed steps,
        # which over-estimate the step time when --steps is tiny)
        n_more, d_more = 0, 0.0
        est = dt / args.steps
        while d_more < args.sustain:
            chunk = max(int(np.ceil((args.sustain - d_more) / est * 1.02)), 1)
            barrier()
            t1 = time.perf_counter()
            for _ in range(chunk):
                one_step()
            barrier()
            ts = torch.tensor([time.perf_counter() - t1], dtype=torch.float64)
            if dist.is_initialized():
                dist.all_reduce(ts, op=dist.ReduceOp.MAX)      # identical on every rank: all ranks leave the loop together
            d_more += float(ts.item())
            n_more += chunk
            est = d_more / n_more
        sustained = {"value": round(world * B * micro * n_more / d_more, 3), "unit": "frames/s", "steps": n_more,
                     "seconds": round(d_more, 2), "ms_per_step": round(d_more / n_more * 1e3, 3),
                     "note": "continuation of the timed region on the same state; `value` above is the --steps window only"}

    # The library's default of two branch streams, beside the headline: `value` / `roofline` are measured on ONE stream so that a
    # per-launch duration is that of a kernel owning the GPU; this window repeats the same step with the vertical encoder on a side
    # stream (bit-identical results: tests/test_fullsize_gpu.py::test_two_stream_branches_match_one_stream_at_batch32).
    two_streams = None
    if args.sustain > 0 and not args.graph and not F_.TWO_STREAMS and os.environ.get("HUPR_ONE_STREAM", "0") != "1":
        F_.TWO_STREAMS = True
        try:
            for _ in range(3):
                one_step()
            n2 = max(int(np.ceil(1.5 / (dt / args.steps))), 10)
            barrier()
            t1 = time.perf_counter()
            for _ in range(n2):
                one_step()
            barrier()
            ts = torch.tensor([time.perf_counter() - t1], dtype=torch.float64)
            if dist.is_initialized():
                dist.all_reduce(ts, op=dist.ReduceOp.MAX)
            d2 = float(ts.item())
            two_streams = {"value": round(world * B * micro * n2 / d2, 3), "unit": "frames/s", "steps": n2, "ms_per_step": round(d2 / n2 * 1e3, 3),
                           "note": "same step, functional.TWO_STREAMS = True (python bench.py --two-streams times the whole run this way)"}
        finally:
            F_.TWO_STREAMS = False

    # N > 1: the line checks itself.  (i) the live native communicator must span all ranks; (ii) per-bucket
    # all-reduce durations and the exposed (non-overlapped) tail from HIP events on the communication stream, a few eager steps
    # after the timed region; (iii) the SAME step with the exchange switched off — every rank an independent replica — as the
    # single-rank rate measured in this very invocation, so that efficiency = value / (N x that) needs no second run.
    scaling_check = None
    if world > 1 or (eng.buckets.active and not args.strong):      # (one rank with HUPR_FORCE_ALLREDUCE=1: the same code path on a 1-GPU box)
        if rccl_ranks is not None and rccl_ranks != world:
            raise SystemExit("bench.py: the native RCCL communicator spans %r ranks, the job has %d" % (rccl_ranks, world))
        if not args.graph:
            eng.buckets.enable_timing(True)
            for _ in range(5):
                one_step()
            rep = eng.buckets.timing_report()
            eng.buckets.enable_timing(False)
            reps = [rep]
            if dist.is_initialized():
                reps = [None] * world
                dist.all_gather_object(reps, rep)
            # (iii) exchange off: rank-local replicas (weights drift apart from here on — this is the last thing the engine does)
            was_active, eng.buckets.active = eng.buckets.active, False
            for _ in range(3):
                one_step()
            n1 = max(int(np.ceil(1.5 / (dt / args.steps))), 10)
            barrier()
            t1 = time.perf_counter()
            for _ in range(n1):
                one_step()
            barrier()
            ts = torch.tensor([time.perf_counter() - t1], dtype=torch.float64)
            if dist.is_initialized():
                dist.all_reduce(ts, op=dist.ReduceOp.MAX)
            eng.buckets.active = was_active
            single = B * micro * n1 / float(ts.item())            # per-rank rate of N concurrent independent replicas
            tails = [r["exposed_tail_us"] for r in reps if r and r.get("exposed_tail_us") is not None]
            scaling_check = {"rccl_ranks": rccl_ranks, "transport": transport,
                             "buckets_rank0": reps[0]["buckets"] if reps[0] else None,
                             "exposed_tail_us_max_over_ranks": max(tails) if tails else None,
                             "single_rank_frames_per_s": round(single, 3),
                             "single_rank_note": "the same step with the exchange switched off, all %d ranks running as independent "
                                                 "replicas at once (same invocation, same boxes, same clocks)" % world,
                             "efficiency_vs_single_rank": round((world * B * micro * args.steps / dt) / (world * single), 4),
                             "mode": "strong (global batch %d, %d micro-batches per rank)" % (STRONG_GLOBAL_BATCH, micro) if args.strong else "weak"}

    if args.graph:      # roofline probe on a few eager steps (events cannot be read back from inside a graph replay)
        eng._graph = None
        F_.CONV_PROBE = conv_probe_factory(probe_events)
        for _ in range(3):
            one_step()
        F_.CONV_PROBE = None
        barrier()

    launch_census = None
    if not args.graph and not args.no_launch_census and not args.no_probes:
        # EVERY rank runs the census' two steps (with an exchange step they contain collectives: rank 0 alone would wait for its
        # peers forever); only rank 0 traces them
        if rank == 0:
            launch_census = count_device_activities(one_step)
        else:
            one_step()
            one_step()
        barrier()
    fft_roof = parity = None
    if rank == 0 and not args.no_probes:
        # FFT chain on its own (HBM-bound): the step's 2 x B*G sensor-frames, HIP events on the launch stream.  Primary entry =
        # the variant the timed step ran (a1 + a2 + the elevation mean of a3 fused: 1 048 576 B algorithmic per sensor-frame);
        # `loader_variant` = the reference-shaped hand-over (a1 + a2: 2 883 584 B per sensor-frame) for comparison.
        def fft_time(fn):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            fn(adc_h)
            ev[0].record()
            for _ in range(10):
                fn(adc_h)
                fn(adc_v)
            ev[1].record()
            torch.cuda.synchronize()
            return ev[0].elapsed_time(ev[1]) * 1e-3 / (20 * B * G)

        trash = torch.empty(1 << 28, dtype=torch.float32, device=dev)      # 1 GiB: four Infinity Caches

        def fft_time_cold(fn):
            """The two sensors' calls back to back with 1 GiB of unrelated traffic in front of the pair, as inside the step (~25 GB
            between two pairs): inputs never in the 256 MB Infinity Cache.  HIP events around the pair, median of 6, per call = half
            (events around ONE call also count ~8 us of launch floor — an empty-bodied kernel measures 13 us that way — which the
            step's back-to-back launches do not pay; that figure is kept as `single_call`)."""
            pair, single = [], []
            for i in range(12):
                trash.fill_(float(i))
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                ev[0].record()
                if i % 2 == 0:
                    fn(adc_h)
                    fn(adc_v)
                else:
                    fn(adc_h if i % 4 == 1 else adc_v)
                ev[1].record()
                torch.cuda.synchronize()
                (pair if i % 2 == 0 else single).append(ev[0].elapsed_time(ev[1]) * 1e-3 / (2 if i % 2 == 0 else 1))
            return sorted(pair)[len(pair) // 2] / (B * G), sorted(single)[len(single) // 2] / (B * G)

        try:      # HBM bytes per sensor-frame of the same kernels, measured offline with rocprofv3 --pmc (profiles/)
            fft_pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_fft.json")))
        except Exception:      # noqa: BLE001
            fft_pmc = {}

        def fft_obj(per_sf, nbytes, kernel, per_sf_cold, pmc_key):
            # `achieved` / `frac` = the COLD figure: one call per sensor behind 1 GiB of unrelated traffic, which
            # is how the training step sees the chain; the back-to-back figure of rounds 1-3 (part of the int16 cube still in the
            # 256 MB Infinity Cache) is kept as the `warm` sub-object.
            per_sf_cold, per_sf_single = per_sf_cold
            gbs, cold = nbytes / per_sf / 1e9, nbytes / per_sf_cold / 1e9
            return {"bound": "hbm", "kernel": kernel, "achieved": round(cold, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(cold / PEAK_HBM_GBS, 4), "algorithmic_bytes_per_sensor_frame": nbytes,
                    "traffic": fft_pmc.get(pmc_key, {}).get("traffic_bytes_per_sensor_frame"),
                    "traffic_note": "HBM bytes per sensor-frame, 2 x FETCH_SIZE + WRITE_SIZE of both kernels (profiles/pmc_fft.json)",
                    "sensor_frames_per_s": round(1.0 / per_sf_cold, 1), "share_of_step_ms": round(per_sf_cold * 2 * B * G * 1e3, 3),
                    "measured": "cold: the two sensors' calls back to back with 1 GiB of unrelated traffic in front of the pair (how the "
                                "step sees them: no Infinity-Cache hits on the int16 cubes), HIP events around the pair, median of 6, per "
                                "call = half; share_of_step_ms uses this figure",
                    "single_call": {"achieved": round(nbytes / per_sf_single / 1e9, 1), "frac": round(nbytes / per_sf_single / 1e9 / PEAK_HBM_GBS, 4),
                                    "note": "events around ONE cold call (the round-4 interim figure): includes ~8 us of launch floor"},
                    "warm": {"achieved": round(gbs, 1), "frac": round(gbs / PEAK_HBM_GBS, 4),
                             "sensor_frames_per_s": round(1.0 / per_sf, 1),
                             "note": "20 back-to-back calls alternating the two sensors' cubes (the rounds-1-3 headline; Infinity-Cache-assisted)"}}
        fused_mean = fft_obj(fft_time(fft_chain_loader_means), FFT_MEANS_BYTES_PER_SF,
                             "hupr_k_doppler_range + hupr_k_angle<loader + elevation mean> (FFT chain, Normalize, HuPRNet's elevation mean)",
                             fft_time_cold(fft_chain_loader_means), "fused_mean")
        loader = fft_obj(fft_time(fft_chain_loader), FFT_LOADER_BYTES_PER_SF,
                         "hupr_k_doppler_range + hupr_k_angle<loader> (FFT chain fused with the loader glue)",
                         fft_time_cold(fft_chain_loader), "loader")
        del trash
        fft_roof = dict(fused_mean if fused_step else loader)
        fft_roof["loader_variant" if fused_step else "fused_mean_variant"] = loader if fused_step else fused_mean
    attn_roof = None
    if rank == 0 and args.dtype == "bf16" and not args.no_probes:
        # MSCSA level-1 attention (C = 64, N = 4096: 88 % of the attention flops) at this batch, kernels alone through the C ABI:
        # forward 4 N^2 C flops per sample, backward (prep + dQ + dK/dV) 10 N^2 C algorithmic; HIP events on the launch stream
        L_, rt_ = F_.rt.lib(), F_.rt
        Na, Ca = 4096, 64
        ga = torch.Generator(device=dev).manual_seed(11)
        ka, qa, va = (torch.randn(B, Na, Ca, device=dev, generator=ga) * s_ for s_ in (0.5, 0.5, 1.0))
        kb_, qb_, vb_ = ka.bfloat16(), qa.bfloat16(), va.bfloat16()
        g32 = torch.randn(B, Na, Ca, device=dev, generator=ga)
        gb_ = g32.bfloat16()
        oa, la = torch.empty(B, Na, Ca, device=dev), torch.empty(B, Na, device=dev)
        dka, dqa, dva = (torch.empty(B, Na, Ca, device=dev) for _ in range(3))
        sca = torch.empty(B, Na, device=dev)
        qs_ = F_.QS_ATTN                                      # what the step's fused MSCSA levels run (round 5: query operand pre-scaled by log2 e)
        if qs_:
            qb_ = (qa * 1.4426950408889634).bfloat16()
            a_fwd = lambda: rt_.check(L_.hupr_attn_fwd_bf16in_ld_ws_qs(rt_.ptr(kb_), Ca, rt_.ptr(qb_), Ca, rt_.ptr(vb_), rt_.ptr(va), rt_.ptr(oa), rt_.ptr(la),      # noqa: E731
                                                                       None, 0, B, Na, Ca, None, 0, rt_.stream()))
            a_bwd = lambda: rt_.check(L_.hupr_attn_bwd_bf16in_ld_qs(rt_.ptr(kb_), Ca, rt_.ptr(qb_), Ca, rt_.ptr(vb_), rt_.ptr(gb_), Ca, rt_.ptr(va), rt_.ptr(oa),      # noqa: E731
                                                                    rt_.ptr(g32), rt_.ptr(la), rt_.ptr(dka), Ca, rt_.ptr(dqa), Ca, rt_.ptr(dva), rt_.ptr(sca),
                                                                    B, Na, Ca, 1, 0, rt_.stream()))
        else:
            a_fwd = lambda: rt_.check(L_.hupr_attn_fwd_bf16in(rt_.ptr(kb_), rt_.ptr(qb_), rt_.ptr(vb_), rt_.ptr(va), rt_.ptr(oa), rt_.ptr(la), B, Na, Ca, rt_.stream()))      # noqa: E731
            a_bwd = lambda: rt_.check(L_.hupr_attn_bwd_bf16in(rt_.ptr(kb_), rt_.ptr(qb_), rt_.ptr(vb_), rt_.ptr(gb_), rt_.ptr(va), rt_.ptr(oa), rt_.ptr(g32), rt_.ptr(la),      # noqa: E731
                                                              rt_.ptr(dka), rt_.ptr(dqa), rt_.ptr(dva), rt_.ptr(sca), B, Na, Ca, 1, rt_.stream()))

        def a_time(fn, n=10):
            for _ in range(3):
                fn()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(n):
                fn()
            ev[1].record()
            torch.cuda.synchronize()
            return ev[0].elapsed_time(ev[1]) * 1e-3 / n
        tf, tb = a_time(a_fwd), a_time(a_bwd)
        ffl, bfl = 4.0 * Na * Na * Ca * B, 10.0 * Na * Na * Ca * B
        # ... and the same launches INSIDE the step (five steps after the timed region; HIP events around each level-1 attention of the
        # fused MSCSA node): back-to-back copies of one matrix-heavy kernel run at the chip's power limit, in the step the kernel
        # follows lighter ones — the in-step duration is what the frames/s above contains (rocprofv3: profiles/r05_bench_kernels.md)
        in_step = None
        if not args.graph and world == 1:            # (the step contains the exchange: one rank alone cannot run extra steps)
            a_ev = {"fwd": [], "bwd": [], "bwd_level": []}

            def a_probe(kind, b_, n_, c_):
                if n_ == Na and c_ == Ca and b_ == B:
                    pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    a_ev[kind].append(pair)
                    return pair
                return None
            F_.ATTN_PROBE = a_probe
            try:
                for _ in range(5):
                    one_step()
                torch.cuda.synchronize()
            finally:
                F_.ATTN_PROBE = None
            if a_ev["fwd"] and (a_ev["bwd"] or a_ev["bwd_level"]):
                tfi = float(np.mean([s_.elapsed_time(e_) for s_, e_ in a_ev["fwd"]])) * 1e-3
                if a_ev["bwd_level"]:          # the level's four backward passes as one call (batched row-sum and dQ launches): per attention
                    tbi = float(np.mean([s_.elapsed_time(e_) for s_, e_ in a_ev["bwd_level"]])) * 1e-3 / 4.0
                    a_ev["bwd"] = a_ev["bwd_level"] * 4
                else:
                    tbi = float(np.mean([s_.elapsed_time(e_) for s_, e_ in a_ev["bwd"]])) * 1e-3
                in_step = {"forward": {"us": round(tfi * 1e6, 1), "achieved": round(ffl / tfi / 1e12, 1), "frac": round(ffl / tfi / 1e12 / peak, 4),
                                       "launches": len(a_ev["fwd"])},
                           "backward": {"us": round(tbi * 1e6, 1), "achieved": round(bfl / tbi / 1e12, 1), "frac": round(bfl / tbi / 1e12 / peak, 4),
                                        "launches": len(a_ev["bwd"])},
                           "frac": round((ffl + bfl) / (tfi + tbi) / 1e12 / peak, 4),
                           "how": "HIP events around every level-1 attention of 5 steps after the timed region (forward: one launch; "
                                  "backward: prep + dQ + dK/dV launches and the gaps between them)"}
        attn_roof = {"bound": "mfma", "kernel": "hupr_k_attn_fwd_pp64 / hupr_k_attn_bwd_dq + hupr_k_attn_bwd_dkv512%s (MSCSA level 1: C = 64, N = 4096, B = %d)" % (" <QS>" if qs_ else "", B),
                     "forward": {"us": round(tf * 1e6, 1), "achieved": round(ffl / tf / 1e12, 1), "frac": round(ffl / tf / 1e12 / peak, 4)},
                     "backward": {"us": round(tb * 1e6, 1), "achieved": round(bfl / tb / 1e12, 1), "frac": round(bfl / tb / 1e12 / peak, 4),
                                  "note": "algorithmic 10 N^2 C; the two kernels execute 14 N^2 C (S and dP recomputed in each)"},
                     "achieved": round((ffl + bfl) / (tf + tb) / 1e12, 1), "peak": peak, "unit": "TFLOP/s",
                     "frac": round((ffl + bfl) / (tf + tb) / 1e12 / peak, 4),
                     "how": "kernels alone, 10 back-to-back launches through the C ABI", "in_step": in_step,
                     "limiter": "non-MFMA instruction issue beside a power-paced matrix pipe: the forward's phases (MFMA 65 us, fragment reads 38, "
                                "soft-max VALU 65-70, stores / DMA / barrier 28) add up instead of overlapping (profiles/r04_attn_ablation.txt, "
                                "profiles/r04_valu_issue_probe.txt)"}
        del ka, qa, va, kb_, qb_, vb_, g32, gb_, oa, la, dka, dqa, dva, sca
    if dist.is_initialized():
        dist.all_reduce(torch.zeros(1))

    if rank == 0 and world == 1 and args.dtype == "bf16" and not args.no_parity_path and not args.strong:
        # the path that meets north_star's 1e-3 heat-map / exact arg-max gate: fp32 matrix pipe, fp32 activations
        eng.close()                  # release the exchange step's communicator (live under HUPR_FORCE_ALLREDUCE)
        del eng
        torch.cuda.empty_cache()
        F_.set_math("f32")
        eng32 = TrainEngine(cfg, device=dev, seed=0)
        ev32 = []
        eng32.train_step_from_adc(adc_h, adc_v, joints, decode="device")
        torch.cuda.synchronize()
        F_.CONV_PROBE = conv_probe_factory(ev32)
        n32 = 4
        t0 = time.perf_counter()
        for _ in range(n32):
            eng32.train_step_from_adc(adc_h, adc_v, joints, decode="device")
        torch.cuda.synchronize()
        d32 = time.perf_counter() - t0
        F_.CONV_PROBE = None
        parity = {"dtype": "f32", "value": round(B * n32 / d32, 2), "unit": "frames/s", "steps": n32,
                  "ms_per_step": round(d32 / n32 * 1e3, 2),
                  "gate": "heat-maps within 1e-3 max-abs of the reference, arg-max identical (tests/test_model_gpu.py)",
                  "roofline": conv_roofline(ev32, B, "f32", PEAK_F32_MFMA_TFLOPS)}
        eng32.close()
        del eng32
        F_.set_math(args.dtype)

    c2 = None
    if rank == 0 and world == 1 and not args.no_c2 and not args.strong:
        # config C2 beside the headline: eval forward, B = 1, the library default of two branch streams inside one hipGraph
        try:
            eng.close()
            del eng
        except NameError:
            pass
        torch.cuda.empty_cache()
        two = F_.TWO_STREAMS
        F_.TWO_STREAMS = os.environ.get("HUPR_ONE_STREAM", "0") != "1"
        try:
            o = bench_inference(args, cfg, dev, rank, 1, peak, steps=250, batch=1, emit=False)
            c2 = {"workload": o["config"]["workload"], "batch": 1, "frames_per_s": o["value"], "latency_ms": o["ms_per_step"],
                  "steps": o["steps"], "launch": o["config"]["launch"], "dtype": o["dtype"], "model_tflops": o["model_tflops"],
                  "roofline": o.get("roofline")}
        finally:
            F_.TWO_STREAMS = two

    if rank == 0:
        frames = world * B * micro * args.steps
        value = frames / dt
        out = {
            "metric": "radar frames/sec (FFT->heatmap fwd+bwd)", "value": round(value, 3), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "C3: mscsa_prgcn training fwd+bwd+Adam with on-GPU FFT preprocess fused into the loader "
                                   "(16 un-cached sensor-frames per sample%s); loss incl. the per-iteration arg-max "
                                   "decodes (on device, results not copied to the host)" % (", elevation mean fused" if fused_step else ""),
                       "batch_per_gpu": B, "micro_batches_per_step": micro, "global_batch": B * micro * world,
                       "parallelism": "dp%d" % world, "model_gflop_per_frame": STEP_GFLOP,
                       "collective": transport,
                       "attention": "bf16",
                       "launch": "hipGraph replay" if args.graph else "eager",
                       "compute_streams": 2 if (F_.TWO_STREAMS and world == 1) else 1},
            "model_tflops": round(value * STEP_GFLOP / 1e3, 2),
            "model_frac_of_mfma_peak": round(value * STEP_GFLOP / 1e3 / world / peak, 4),
            "loss": round(loss_value, 5),
            "launches_per_step": round(launches_native, 1),
            "launches_per_step_note": "library launches counted by the C ABI over the timed region, in THIS run's stream mode (%s): with one compute "
                                      "stream the twelve PReLU slope-gradient sums are one deferred launch (functional.PRELU_DEFER), with two "
                                      "streams they are twelve" % ("two compute streams" if F_.TWO_STREAMS else "one compute stream"),
            "launch_census": launch_census,
            "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 2),
            "host_enqueue_ms_per_step_by_rank": [round(float(x), 2) for x in enq_ranks],
            "host_enqueue_ms_from_idle_gpu": round(host_enqueue_idle_ms, 2),
            "rccl_ranks": rccl_ranks,
            "scaling_check": scaling_check,
            "sustained": sustained,
            "two_streams": two_streams,
            "roofline": conv_roofline(probe_events, B, args.dtype, peak),
            "fft_roofline": fft_roof,
        }
        if attn_roof is not None:
            out["attention_roofline"] = attn_roof
        if parity is not None:
            out["parity_path"] = parity
        if c2 is not None:
            out["c2"] = c2
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
