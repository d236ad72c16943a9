"""GPU (-m gpu): the data-parallel exchange step and the engine features around it, on ONE MI355X.

RCCL refuses two ranks on one device, so the exchange runs as a single-rank communicator (``HUPR_FORCE_ALLREDUCE=1``):
every byte still goes through ``hupr_allreduce_bucket`` (ncclAllReduce on the C ABI's own communicator, enqueued on the
communication stream and joined before Adam), and the sum over one rank must leave every bit unchanged.  The world-size-2
logic (bucket order, broadcast, accumulation) is covered on CPU over gloo in tests/test_distributed_cpu.py.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

from hupr_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(B=4, seed=51):
    from hupr_amd.config_tree import load_config
    cfg = load_config()
    dev = torch.device("cuda", 0)
    G = cfg.DATASET.numGroupFrames
    adc_h = torch.from_numpy(synth.adc_cube_int16(seed, sensor=0, nframes=B * G)).to(dev)
    adc_v = torch.from_numpy(synth.adc_cube_int16(seed, sensor=1, nframes=B * G)).to(dev)
    joints = torch.from_numpy(synth.keypoints(B, seed + 1)).to(dev)
    return cfg, dev, adc_h, adc_v, joints


def _flat(eng):
    return torch.cat([p.detach().flatten() for p in eng.model.parameters()])


def test_rccl_exchange_single_rank_is_bit_transparent(monkeypatch):
    """3 optimisation steps with every gradient bucket pushed through hupr_allreduce_bucket on the communication stream
    == 3 steps without any collective, bit for bit (parameters, BatchNorm statistics, loss)."""
    from hupr_amd import functional as F_
    from hupr_amd.tools.engine import TrainEngine
    saved = F_.TWO_STREAMS
    try:
        F_.set_math("bf16")
        F_.TWO_STREAMS = False
        cfg, dev, adc_h, adc_v, joints = _setup()
        monkeypatch.delenv("HUPR_FORCE_ALLREDUCE", raising=False)
        e0 = TrainEngine(cfg, device=dev, seed=0)
        assert not e0.buckets.active and e0.buckets.transport is None
        monkeypatch.setenv("HUPR_FORCE_ALLREDUCE", "1")
        e1 = TrainEngine(cfg, device=dev, seed=0)
        assert e1.buckets.active and e1.buckets.transport.name.startswith("rccl"), e1.buckets.transport.name
        for _ in range(3):
            l0, _ = e0.train_step_from_adc(adc_h, adc_v, joints)
            l1, _ = e1.train_step_from_adc(adc_h, adc_v, joints)
        torch.cuda.synchronize()
        assert all(b.launched for b in e1.buckets.buckets)
        assert float(l0) == float(l1)
        assert torch.equal(_flat(e0), _flat(e1))
        s0, s1 = e0.model.state_dict(), e1.model.state_dict()
        for k in s0:
            assert torch.equal(s0[k], s1[k]), k
        e1.buckets.transport.close()
    finally:
        F_.TWO_STREAMS = saved
        F_.set_math("f32")


def test_rccl_exchange_at_batch32_with_two_streams_is_bit_transparent(monkeypatch):
    """The regime a data-parallel rank really runs in: B = 32, both encoder branches in flight on two streams AND the bucket
    all-reduces on the communication stream beside the backward kernels.  Two steps that way == two steps on one stream without
    any collective, bit for bit.  (Round 3: at B = 32 kernels of different streams really share the chip, and one kernel turned
    out to return wrong sums beside another stream's convolution — DESIGN.md section 7; the B = 4 test above never saw it.)"""
    import numpy as np
    import pose_fit
    from hupr_amd import functional as F_
    from hupr_amd.config_tree import load_config
    from hupr_amd.tools.engine import TrainEngine
    saved = F_.TWO_STREAMS
    try:
        F_.set_math("bf16")
        cfg = load_config()
        dev = torch.device("cuda", 0)
        h, v, joints = pose_fit.scene_batch(32, np.random.default_rng(7), torch.Generator(device=dev).manual_seed(8), dev)
        F_.TWO_STREAMS = False
        monkeypatch.delenv("HUPR_FORCE_ALLREDUCE", raising=False)
        e0 = TrainEngine(cfg, device=dev, seed=0, lr=2e-4)
        for _ in range(2):
            l0, _ = e0.train_step(h, v, joints)
        torch.cuda.synchronize()
        ref = {k: t.detach().clone() for k, t in e0.model.state_dict().items()}
        e0.close()
        F_.TWO_STREAMS = True
        monkeypatch.setenv("HUPR_FORCE_ALLREDUCE", "1")
        for attempt in range(2):
            e1 = TrainEngine(cfg, device=dev, seed=0, lr=2e-4)
            assert e1.buckets.active and e1.buckets.transport.name.startswith("rccl"), e1.buckets.transport.name
            for _ in range(2):
                l1, _ = e1.train_step(h, v, joints)
            torch.cuda.synchronize()
            assert all(b.launched for b in e1.buckets.buckets)
            assert float(l0.detach()) == float(l1.detach())
            bad = [k for k, t in e1.model.state_dict().items() if not torch.equal(t, ref[k])]
            assert not bad, (attempt, bad[:8])
            e1.close()
    finally:
        F_.TWO_STREAMS = saved
        F_.set_math("f32")


def test_graph_captures_the_exchange_step(monkeypatch):
    """The data-parallel step as ONE hipGraph: the fork to the communication stream, ncclAllReduce and the join before Adam
    are captured; 2 eager + 1 warm-up + 2 replays == 5 eager steps."""
    from hupr_amd import functional as F_
    from hupr_amd.tools.engine import TrainEngine
    saved = F_.TWO_STREAMS
    try:
        F_.set_math("bf16")
        F_.TWO_STREAMS = False
        monkeypatch.setenv("HUPR_FORCE_ALLREDUCE", "1")
        cfg, dev, adc_h, adc_v, joints = _setup(seed=61)
        e1 = TrainEngine(cfg, device=dev, seed=0)
        for _ in range(5):
            l1, _ = e1.train_step_from_adc(adc_h, adc_v, joints)
        e2 = TrainEngine(cfg, device=dev, seed=0)
        for _ in range(2):
            e2.train_step_from_adc(adc_h, adc_v, joints)
        hv = e2.preprocess(adc_h, adc_v)
        e2.infer(*hv)                                   # fills the host-side inference constants with the weights of step 2
        e2.capture(adc_h, adc_v, joints, warmup=1, decode="device")
        for _ in range(2):
            l2, _ = e2.train_step_from_adc(adc_h, adc_v, joints)
        torch.cuda.synchronize()
        # the replayed graph's Adam node moves the parameters behind every host-side cache; an evaluation after
        # replayed steps must see the weights of step 5, not the constants cached at step 2
        from hupr_amd.models import HuPRNet
        got = e2.infer(*hv)
        fresh = HuPRNet(cfg).to(dev).eval()
        fresh.load_state_dict(e2.model.state_dict())
        F_.invalidate_packed()
        with torch.no_grad():
            want = fresh(*hv)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
        p1, p2 = _flat(e1), _flat(e2)
        assert torch.isfinite(p2).all()
        rel = ((p1 - p2).norm() / p1.norm()).item()
        assert rel <= 1e-5, rel
        assert abs(float(l1) - float(l2)) <= 1e-4 * abs(float(l1))
        # guards: host joints / a loss-weight schedule cannot be captured
        e3 = TrainEngine(cfg, device=dev, seed=0)
        with pytest.raises(RuntimeError, match="resident on the GPU"):
            e3.capture(adc_h, adc_v, joints.cpu())
    finally:
        F_.TWO_STREAMS = saved
        F_.set_math("f32")


def test_accumulated_micro_batches_equal_one_step():
    """bench.py --strong: two identical micro-batches accumulated into one optimiser step give exactly the parameters of
    one step on that micro-batch ((g + g) / 2 == g in binary floating point); BatchNorm statistics saw two batches."""
    from hupr_amd import functional as F_
    from hupr_amd.tools.engine import TrainEngine
    try:
        F_.set_math("bf16")
        cfg, dev, adc_h, adc_v, joints = _setup(seed=71)
        e1 = TrainEngine(cfg, device=dev, seed=0)
        e1.train_step_from_adc(adc_h, adc_v, joints)
        e2 = TrainEngine(cfg, device=dev, seed=0)
        e2.train_step_accumulated([(adc_h, adc_v, joints)] * 2)
        torch.cuda.synchronize()
        assert torch.equal(_flat(e1), _flat(e2))
        nbt = [int(v) for k, v in e2.model.state_dict().items() if k.endswith("num_batches_tracked")]
        assert set(nbt) == {2}
        # and a second accumulated step starts from a clean accumulator
        e1.train_step_from_adc(adc_h, adc_v, joints)
        e2.train_step_accumulated([(adc_h, adc_v, joints)] * 2)
        rel = ((_flat(e1) - _flat(e2)).norm() / _flat(e1).norm()).item()
        assert rel < 1e-4, rel            # BatchNorm running stats differ (2 vs 1 updates) but train-mode outputs do not
    finally:
        F_.set_math("f32")


def test_checkpoint_resume_is_bit_exact(tmp_path):
    """optimizer_state_dict must carry the Adam moments.  train 2 steps -> save -> (fresh engine) load ->
    step 3 lands on the same bits as the uninterrupted run; the saved state has torch.optim.Adam's layout."""
    from hupr_amd import functional as F_
    from hupr_amd.tools.engine import TrainEngine
    try:
        F_.set_math("bf16")
        cfg, dev, adc_h, adc_v, joints = _setup(seed=81)
        e1 = TrainEngine(cfg, device=dev, seed=0)
        for _ in range(2):
            e1.train_step_from_adc(adc_h, adc_v, joints)
        ck = {"model_state_dict": e1.model.state_dict(), "optimizer_state_dict": e1.optimizer.state_dict()}
        torch.save(ck, tmp_path / "checkpoint.pth")
        e1.train_step_from_adc(adc_h, adc_v, joints)
        n_params = sum(1 for _ in e1.model.parameters())
        ck = torch.load(tmp_path / "checkpoint.pth", map_location="cuda")
        st = ck["optimizer_state_dict"]["state"]
        assert len(st) == n_params and all(float(s["step"]) == 2 and s["exp_avg_sq"].abs().sum() > 0 for s in st.values())
        e2 = TrainEngine(cfg, device=dev, seed=123)              # different init on purpose
        e2.model.load_state_dict(ck["model_state_dict"])
        e2.optimizer.load_state_dict(ck["optimizer_state_dict"])
        F_.invalidate_packed()
        e2.train_step_from_adc(adc_h, adc_v, joints)
        torch.cuda.synchronize()
        assert torch.equal(_flat(e1), _flat(e2))
        # the same state drives torch.optim.Adam (the reference's optimiser, tools/base.py:47)
        tad = torch.optim.Adam(e2.model.parameters(), lr=cfg.TRAINING.lr, weight_decay=1e-4)
        tad.load_state_dict(ck["optimizer_state_dict"])
        assert len(tad.state) == n_params
    finally:
        F_.set_math("f32")


def _bench(args, env_extra=None, launcher=False):
    env = dict(os.environ)
    env.update(env_extra or {})
    cmd = [sys.executable]
    if launcher:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                "--master-port", "29671"]
    cmd += [os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_single_gpu_carries_every_object():
    out = _bench(["--steps", "3", "--warmup", "1", "--batch", "4", "--no-cpu-baseline"])
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["scaling"] == "weak" and out["dtype"] == "bf16"
    assert out["config"]["parallelism"] == "dp1" and out["config"]["launch"] == "eager"
    assert out["roofline"]["launches"] == 3 * 12 and 0 < out["roofline"]["frac"] < 1
    assert out["fft_roofline"]["bound"] == "hbm" and 0 < out["fft_roofline"]["frac"] < 1 and "loader_variant" in out["fft_roofline"]
    assert 0 < out["fft_roofline"]["warm"]["frac"] < 1                  # `frac` itself is the cold figure since round 4
    assert out["attention_roofline"]["bound"] == "mfma" and 0 < out["attention_roofline"]["frac"] < 1
    assert out["parity_path"]["dtype"] == "f32" and out["parity_path"]["roofline"]["peak"] == 157.3
    assert "arg-max" in out["config"]["workload"]
    # round 5: the step's launch count (native counter) and the bounds of the single-sample forward next to its latency
    assert 100 < out["launches_per_step"] < 1000
    c2r = out["c2"]["roofline"]
    assert c2r["bound"] == "launch" and 20 < c2r["launches_per_frame"] < 400 and 0 < c2r["frac"] < 1 and 0 < c2r["frac_of_launch_floor"] <= 1.0
    assert c2r["bounds_us"]["launch_floor_us"] == round(c2r["launches_per_frame"] * 1.5, 1)
    # SURVEY 8(d)'s algorithmic bytes, not the implementation's traffic
    assert out["fft_roofline"]["algorithmic_bytes_per_sensor_frame"] == 1048576
    assert out["fft_roofline"]["loader_variant"]["algorithmic_bytes_per_sensor_frame"] == 2883584
    # the line is self-sufficient: a sustained continuation and config C2 beside the burst number
    assert out["sustained"]["seconds"] >= 3.0 and out["sustained"]["steps"] >= 1 and out["sustained"]["value"] > 0
    assert out["c2"]["batch"] == 1 and out["c2"]["steps"] == 250 and 0 < out["c2"]["latency_ms"] < 50
    assert out["rccl_ranks"] is None and len(out["host_enqueue_ms_per_step_by_rank"]) == 1


def test_bench_under_launcher_uses_rccl_and_graph():
    """One rank started the way the driver starts N: gloo control plane + the C ABI's RCCL communicator for the buckets;
    --graph replays the captured data-parallel step."""
    out = _bench(["--gpus", "1", "--steps", "3", "--warmup", "2", "--batch", "4", "--no-cpu-baseline", "--no-parity-path", "--graph"],
                 env_extra={"HUPR_FORCE_ALLREDUCE": "1"}, launcher=True)
    assert out["n_gpus"] == 1 and out["config"]["collective"].startswith("rccl")
    assert out["config"]["launch"] == "hipGraph replay" and out["roofline"]["launches"] == 3 * 12
    assert out["rccl_ranks"] == 1                                  # as the live communicator reports it (ncclCommCount)


def test_bench_line_checks_its_own_exchange_step():
    """with an exchange step in the job the line carries `scaling_check` — the live communicator's rank count
    (asserted against the job inside bench.py), per-bucket all-reduce durations from HIP events on the communication stream, the
    exposed tail at the join, and the same step with the exchange switched off as the single-rank rate of the same invocation.  Here:
    one rank with a forced (1-rank) RCCL all-reduce — the code path N > 1 takes."""
    out = _bench(["--gpus", "1", "--steps", "3", "--warmup", "2", "--batch", "4", "--no-cpu-baseline", "--no-parity-path", "--no-c2",
                  "--sustain", "0"], env_extra={"HUPR_FORCE_ALLREDUCE": "1"}, launcher=True)
    sc = out["scaling_check"]
    assert sc is not None and sc["rccl_ranks"] == 1 and sc["transport"].startswith("rccl")
    assert len(sc["buckets_rank0"]) >= 3 and all(b["allreduce_us"] > 0 and b["mbytes"] > 0 for b in sc["buckets_rank0"])
    assert abs(sum(b["mbytes"] for b in sc["buckets_rank0"]) - 35.54 * 4) < 1.0          # every parameter is in exactly one bucket
    assert sc["exposed_tail_us_max_over_ranks"] is not None and sc["exposed_tail_us_max_over_ranks"] >= 0.0
    assert sc["single_rank_frames_per_s"] > 0 and 0.5 < sc["efficiency_vs_single_rank"] < 1.5
    assert out["launches_per_step"] > 100


def test_bench_refuses_more_gpus_than_visible():
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "visible" in r.stderr and "{" not in r.stdout


def test_bench_strong_scaling_accumulates_micro_batches():
    out = _bench(["--steps", "2", "--warmup", "1", "--batch", "32", "--strong", "--no-cpu-baseline"])
    assert out["scaling"] == "strong" and out["config"]["global_batch"] == 256 and out["config"]["micro_batches_per_step"] == 8
    assert abs(out["value"] - 256 * 2 / (out["ms_per_step"] * 2e-3)) < 1e-2 * out["value"]


def test_fused_elevation_mean_step_equals_reference_shaped_handover(monkeypatch):
    """TrainEngine.preprocess hands the model the elevation-mean planes by default; three optimisation steps land on exactly the
    parameters of the (B,G,F,2,R,A,E) hand-over (HUPR_NO_FUSED_MEAN=1)."""
    from hupr_amd import functional as F_
    from hupr_amd.tools.engine import TrainEngine
    try:
        F_.set_math("bf16")
        cfg, dev, adc_h, adc_v, joints = _setup(seed=91)
        monkeypatch.delenv("HUPR_NO_FUSED_MEAN", raising=False)
        e1 = TrainEngine(cfg, device=dev, seed=0)
        assert e1.fuse_elevation_mean and e1.preprocess(adc_h, adc_v)[0].shape == (4, 8, 16, 64, 64)
        monkeypatch.setenv("HUPR_NO_FUSED_MEAN", "1")
        e2 = TrainEngine(cfg, device=dev, seed=0)
        assert not e2.fuse_elevation_mean and e2.preprocess(adc_h, adc_v)[0].shape == (4, 8, 8, 2, 64, 64, 8)
        for _ in range(3):
            l1, _ = e1.train_step_from_adc(adc_h, adc_v, joints)
            l2, _ = e2.train_step_from_adc(adc_h, adc_v, joints)
        torch.cuda.synchronize()
        assert float(l1.detach()) == float(l2.detach()) and torch.equal(_flat(e1), _flat(e2))
    finally:
        F_.set_math("f32")


# ---- two real ranks: run only where >= 2 GPUs are visible (the first multi-GPU box validates hupr_comm_init_rank(n > 1)) ----
needs_two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 visible GPUs")


@needs_two_gpus
def test_bench_two_ranks_over_rccl():
    """`bench.py --gpus 2` spawns two ranks; the native communicator must report 2 ranks, every rank its enqueue time."""
    out = _bench(["--gpus", "2", "--steps", "3", "--warmup", "2", "--batch", "4", "--no-cpu-baseline", "--sustain", "0"])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["config"]["global_batch"] == 8
    assert out["config"]["collective"].startswith("rccl") and out["rccl_ranks"] == 2
    assert len(out["host_enqueue_ms_per_step_by_rank"]) == 2 and out["value"] > 0


def _dp2_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from hupr_amd import functional as F_
    from hupr_amd.config_tree import load_config
    from hupr_amd.tools.engine import TrainEngine
    torch.cuda.set_device(rank)
    dist.init_process_group("cpu:gloo,cuda:nccl", rank=rank, world_size=world)
    F_.set_math("bf16")
    F_.TWO_STREAMS = False
    cfg = load_config()
    dev = torch.device("cuda", rank)
    eng = TrainEngine(cfg, device=dev, seed=rank)                  # different init per rank: the broadcast must fix it
    assert eng.buckets.transport.name.startswith("rccl") and eng.buckets.transport.ranks() == (world, rank)
    h, v = (torch.from_numpy(t).to(dev) for t in synth.model_inputs(2, 300 + rank))
    joints = torch.from_numpy(synth.keypoints(2, 310 + rank))
    for _ in range(2):
        eng.train_step(h, v, joints)
    torch.cuda.synchronize()
    torch.save(_flat(eng).cpu(), os.path.join(out_dir, "p%d.pt" % rank))
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


@needs_two_gpus
def test_two_rank_step_equals_the_single_process_accumulated_step(tmp_path):
    """Two ranks x one micro-batch each, gradients summed by ncclAllReduce on the communication stream, == one process
    accumulating the same two micro-batches (BatchNorm statistics are per micro-batch in both): identical parameters on both
    ranks, and equal to the single-process result (a + b is the same fp32 number whichever device adds it)."""
    import torch.multiprocessing as mp
    from hupr_amd import functional as F_
    from hupr_amd.config_tree import load_config
    from hupr_amd.tools.engine import TrainEngine
    mp.spawn(_dp2_worker, args=(2, 29683, str(tmp_path)), nprocs=2, join=True)
    p0, p1 = torch.load(tmp_path / "p0.pt"), torch.load(tmp_path / "p1.pt")
    assert torch.equal(p0, p1), "ranks diverged"
    saved = F_.TWO_STREAMS
    try:
        F_.set_math("bf16")
        F_.TWO_STREAMS = False
        dev = torch.device("cuda", 0)
        eng = TrainEngine(load_config(), device=dev, seed=0)      # seed 0 == rank 0's init, which the broadcast distributed
        mb = []
        for r in range(2):
            h, v = (torch.from_numpy(t).to(dev) for t in synth.model_inputs(2, 300 + r))
            mb.append((h, v, torch.from_numpy(synth.keypoints(2, 310 + r))))
        for _ in range(2):
            eng.train_step_accumulated(mb, from_adc=False)
        torch.cuda.synchronize()
        ref = _flat(eng).cpu()
    finally:
        F_.TWO_STREAMS = saved
        F_.set_math("f32")
    rel = ((p0 - ref).norm() / ref.norm()).item()
    assert rel < 1e-6, rel


def test_gradient_slots_written_in_place_and_unzeroed_buckets_are_bit_transparent(monkeypatch):
    """Round 5: (i) the four 1x1 projection weights of an MSCSA level and map sit next to each other in their flat bucket
    (HuPRNet.gradient_groups) and their fused (4C, C) weight gradient is written straight into the four slots; (ii) a bucket whose
    slots were all written by kernels in the previous pass is not zero-filled again.  Four optimisation steps land on exactly the
    parameters of the engine with HUPR_ZERO_GRADS=1 (unconditional fills); after the first pass no fill is issued any more, every
    slot is still written in every pass; a slot that stops receiving gradients is cleared instead of keeping stale values."""
    from hupr_amd import functional as F_
    from hupr_amd.tools.engine import TrainEngine
    try:
        F_.set_math("bf16")
        cfg, dev, adc_h, adc_v, joints = _setup(seed=77)
        monkeypatch.setenv("HUPR_ZERO_GRADS", "1")
        e0 = TrainEngine(cfg, device=dev, seed=0)
        monkeypatch.delenv("HUPR_ZERO_GRADS")
        e1 = TrainEngine(cfg, device=dev, seed=0)
        assert e0.buckets.always_zero and not e1.buckets.always_zero
        # adjacency: every group is consecutive inside one bucket, in the group's order
        pos = {id(p): (bi, i) for bi, b in enumerate(e1.buckets.buckets) for i, p in enumerate(b.params)}
        groups = e1.model.gradient_groups()
        assert len(groups) == 6
        for grp in groups:
            where = [pos[id(p)] for p in grp]
            assert all(w[0] == where[0][0] and w[1] == where[0][1] + j for j, w in enumerate(where)), where
        for step in range(4):
            l0, _ = e0.train_step_from_adc(adc_h, adc_v, joints)
            l1, _ = e1.train_step_from_adc(adc_h, adc_v, joints)
            assert all(all(b.written) and b.clean for b in e1.buckets.buckets)          # every slot written by a kernel, every pass
            # every parameter counted exactly once (the engine visits a directly written parameter's accumulation node too: rounds 1-4
            # ended a pass at pending == -len(params), i.e. a bucket's exchange would have started half-way through its gradients)
            assert all(b.pending == 0 and all(b.arrived) for b in e0.buckets.buckets + e1.buckets.buckets)
            assert [b.zeroed for b in e1.buckets.buckets] == [step == 0] * len(e1.buckets.buckets)
        torch.cuda.synchronize()
        assert float(l0) == float(l1) and torch.equal(_flat(e0), _flat(e1))
        for b0, b1 in zip(e0.buckets.buckets, e1.buckets.buckets):
            assert torch.equal(b0.flat_grad, b1.flat_grad)
        # a pass in which a slot receives nothing: it must read zero afterwards, not the previous pass's gradient
        e1.buckets.prepare()
        b = e1.buckets.buckets[0]
        assert not b.zeroed and b.flat_grad.abs().max().item() > 0
        b.written = [True] * len(b.params)
        b.written[1] = False
        for bb in e1.buckets.buckets[1:]:
            bb.written = [True] * len(bb.params)
        e1.buckets.finish()
        assert b.views[1].abs().max().item() == 0.0 and b.views[0].abs().max().item() > 0 and not b.clean
        e1.buckets.prepare()
        assert e1.buckets.buckets[0].zeroed                                          # not clean -> filled again
        e0.close(); e1.close()
    finally:
        F_.GRAD_SINK = None
        F_.set_math("f32")
        F_.invalidate_packed()


def test_bucket_exchange_starts_after_its_last_gradient(monkeypatch):
    """A bucket's all-reduce may be enqueued only once EVERY gradient of the bucket has been written (round 5: rounds 1-4 counted a
    directly written parameter twice — once when its kernel was enqueued, once when the autograd engine visited its accumulation node —
    and would have launched at the half-way point; invisible with one rank).  One rank, forced collective: at each launch all
    parameters of the bucket have arrived, each bucket launches exactly once per pass, in the same order every pass."""
    from hupr_amd import functional as F_
    from hupr_amd.tools.engine import TrainEngine
    try:
        F_.set_math("bf16")
        monkeypatch.setenv("HUPR_FORCE_ALLREDUCE", "1")
        cfg, dev, adc_h, adc_v, joints = _setup(seed=83)
        eng = TrainEngine(cfg, device=dev, seed=0)
        seen = []
        orig = eng.buckets._launch

        def launch(b):
            seen.append((eng.buckets.buckets.index(b), sum(b.arrived), len(b.params), sum(b.written)))
            return orig(b)
        eng.buckets._launch = launch
        for _ in range(2):
            seen.clear()
            eng.train_step_from_adc(adc_h, adc_v, joints)
            # (each bucket exactly once; the order is the order in which their last gradients land — the same on every rank)
            assert sorted(s[0] for s in seen) == list(range(len(eng.buckets.buckets))), seen
            order = [s[0] for s in seen] if _ == 0 else order
            assert [s[0] for s in seen] == order
            assert all(a == n and w == n for _, a, n, w in seen), seen
        torch.cuda.synchronize()
        eng.close()
    finally:
        F_.GRAD_SINK = None
        F_.set_math("f32")
        F_.invalidate_packed()


@pytest.mark.gpu
def test_deferred_slope_gradient_sums_are_bit_transparent(monkeypatch):
    """The twelve PReLU slope gradients of a step: each backward leaves its per-workgroup partial sums, the gradient sink launches ONE
    multi-item final sum when the first bucket completes after them (in front of that bucket's exchange) instead of twelve
    single-workgroup launches.  Same sums: three optimisation steps — with the exchange step in the job — land on exactly the
    parameters of the engine that sums at once; 11 launches per step less; nothing left pending after a pass."""
    from hupr_amd import functional as F_
    from hupr_amd.tools.engine import TrainEngine
    monkeypatch.setenv("HUPR_FORCE_ALLREDUCE", "1")
    saved = (F_.PRELU_DEFER, F_.TWO_STREAMS)
    try:
        F_.set_math("bf16")
        F_.TWO_STREAMS = False
        cfg, dev, adc_h, adc_v, joints = _setup(seed=78)
        e0 = TrainEngine(cfg, device=dev, seed=0)
        e1 = TrainEngine(cfg, device=dev, seed=0)
        counts = []
        for step in range(3):
            F_.PRELU_DEFER = False
            n0 = F_.rt.lib().hupr_launch_count()
            l0, _ = e0.train_step_from_adc(adc_h, adc_v, joints)
            n1 = F_.rt.lib().hupr_launch_count()
            F_.PRELU_DEFER = True
            l1, _ = e1.train_step_from_adc(adc_h, adc_v, joints)
            n2 = F_.rt.lib().hupr_launch_count()
            counts.append((n1 - n0) - (n2 - n1))
            assert not e1.buckets._deferred and all(b.pending == 0 and all(b.arrived) for b in e1.buckets.buckets)
        torch.cuda.synchronize()
        assert float(l0) == float(l1) and torch.equal(_flat(e0), _flat(e1))
        for b0, b1 in zip(e0.buckets.buckets, e1.buckets.buckets):
            assert torch.equal(b0.flat_grad, b1.flat_grad)
        assert counts[-1] == 11, counts
    finally:
        F_.PRELU_DEFER, F_.TWO_STREAMS = saved
        F_.set_math("f32")


def test_every_sink_writing_kernel_overwrites_its_whole_slot(monkeypatch):
    """``GradientBuckets.prepare`` skips the zero fill of a bucket whose slots were all kernel-written in the previous pass, on
    the assumption that every such kernel OVERWRITES its slot.  HUPR_ZERO_GRADS=poison NaN-fills those buckets instead and ``finish``
    raises if a NaN survives: three steps at B = 4 (one compute stream: the deferred slope sums; then two) must stay finite and land
    on the parameters of the engine that zero-fills unconditionally."""
    from hupr_amd import functional as F_
    from hupr_amd.tools.engine import TrainEngine
    saved = F_.TWO_STREAMS
    try:
        F_.set_math("bf16")
        cfg, dev, adc_h, adc_v, joints = _setup(seed=91)
        for two in (False, True):
            F_.TWO_STREAMS = two
            monkeypatch.setenv("HUPR_ZERO_GRADS", "1")
            e0 = TrainEngine(cfg, device=dev, seed=0)
            monkeypatch.setenv("HUPR_ZERO_GRADS", "poison")
            e1 = TrainEngine(cfg, device=dev, seed=0)
            assert e1.buckets.poison and not e1.buckets.always_zero
            for step in range(3):
                l0, _ = e0.train_step_from_adc(adc_h, adc_v, joints)
                l1, _ = e1.train_step_from_adc(adc_h, adc_v, joints)          # raises inside finish() if a NaN survived
                assert [b.zeroed for b in e1.buckets.buckets] == [step == 0] * len(e1.buckets.buckets)
            torch.cuda.synchronize()
            assert float(l0) == float(l1) and torch.equal(_flat(e0), _flat(e1)), two
            e0.close(); e1.close()
    finally:
        F_.TWO_STREAMS = saved
        F_.GRAD_SINK = None
        F_.set_math("f32")
        F_.invalidate_packed()
