"""Per-call timing of the GEMM-path entry points inside one training step (projections, temporal merges, head, level-0 attention,
PRGCN): every call is bracketed by events and synchronised, so the numbers are kernel times without overlap.
Prints shape arguments, microseconds, and the minimum HBM bytes the call has to move."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hupr_amd import functional as F_, synth, runtime as rt
from hupr_amd.config_tree import load_config
from hupr_amd.tools.engine import TrainEngine
F_.set_math("bf16"); F_.TWO_STREAMS = False
cfg = load_config(); dev = torch.device("cuda", 0)
eng = TrainEngine(cfg, device=dev, seed=0)
B = 32
base = torch.from_numpy(synth.adc_cube_int16(10, sensor=0, nframes=16)).to(dev)
adc_h = base.repeat(16, 1, 1, 1, 1).contiguous(); adc_v = adc_h.clone()
joints = torch.from_numpy(synth.keypoints(B, 20)).to(dev)
for _ in range(2): eng.train_step_from_adc(adc_h, adc_v, joints, decode="device")
torch.cuda.synchronize()
L = rt.lib()
log = []
def wrap(name, describe):
    orig = getattr(L, name)
    def f(*a):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record(); rc = orig(*a); e.record(); torch.cuda.synchronize()
        log.append((name, describe(a), s.elapsed_time(e) * 1e3))
        return rc
    setattr(L, name, f)
wrap("hupr_gemm_bf16", lambda a: "ta%d tb%d M%d N%d K%d batch%d acc%d res%d" % (a[0], a[1], a[5], a[6], a[7], a[11], a[18], a[15] is not None))
wrap("hupr_gemm_f32", lambda a: "ta%d tb%d M%d N%d K%d batch%d" % (a[0], a[1], a[5], a[6], a[7], a[11]))
wrap("hupr_conv_fwd_bf16", lambda a: "B%d in(%d,%d,%d) Ci%d Co%d k(%d,%d,%d)" % (a[5], a[6], a[7], a[8], a[9], a[14], a[17], a[18], a[19]))
wrap("hupr_conv_fwd_bf16_mixed", lambda a: "B%d in(%d,%d,%d) Ci%d Co%d k(%d,%d,%d) xbf%d" % (a[6], a[7], a[8], a[9], a[10], a[15], a[17], a[18], a[19], a[1]))
wrap("hupr_conv_wgrad_bf16", lambda a: "B%d in(%d,%d,%d) Ci%d Co%d k(%d,%d,%d)" % (a[3], a[4], a[5], a[6], a[7], a[12], a[14], a[15], a[16]))
wrap("hupr_conv_wgrad_bf16_mixed", lambda a: "B%d in(%d,%d,%d) Ci%d Co%d k(%d,%d,%d)" % (a[4], a[5], a[6], a[7], a[8], a[13], a[15], a[16], a[17]))
wrap("hupr_tmerge_dgrad_bf16", lambda a: "B%d G%d HW%d Ci%d Co%d" % (a[4], a[5], a[6], a[7], a[8]))
wrap("hupr_tmerge_fwd_stream_bf16", lambda a: "B%d G%d HW%d Ci%d Co%d" % (a[3], a[4], a[5], a[6], a[7]))
wrap("hupr_tmerge_dgrad_stream_bf16", lambda a: "B%d G%d HW%d Ci%d Co%d" % (a[3], a[4], a[5], a[6], a[7]))
wrap("hupr_tmerge_wgrad_stream_bf16", lambda a: "B%d G%d HW%d Ci%d Co%d (incl. the partial reduce)" % (a[3], a[4], a[5], a[6], a[7]))
wrap("hupr_gcn_wx_f32", lambda a: "PRGCN product Bn%d F%d trans%d (dedicated kernel, was hupr_gemm_f32 M1024 N16 K1024 batch32)" % (a[3], a[4], a[6]))
wrap("hupr_gcn_dw_f32", lambda a: "PRGCN weight gradient Bn%d F%d (dedicated kernel, was hupr_gemm_f32 M1024 N1024 K512)" % (a[3], a[4]))
eng.train_step_from_adc(adc_h, adc_v, joints, decode="device")
torch.cuda.synchronize()
agg = collections.OrderedDict()
for n, d, us in log:
    k = n + " " + d
    agg.setdefault(k, [0, 0.0]); agg[k][0] += 1; agg[k][1] += us
tot = sum(v[1] for v in agg.values())
print("GEMM-path calls in one step: %d, %.0f us total" % (len(log), tot))
for k, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%3d x %7.1f us = %7.0f us  %s" % (c, us / c, us, k))
