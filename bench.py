#!/usr/bin/env python3
"""bench.py -- throughput of the detection hot path on MI355X (driver contract: see the task statement).

    python bench.py --gpus N --steps K --warmup W
        N > 1: one rank per GPU over RCCL.  Started by torch.distributed.run (RANK / WORLD_SIZE in the environment) the process is
        one of the ranks; started plainly it first becomes the parent of N fresh rank processes (single_shot_detection_amd/launch.py,
        the role of bf/training/helpers.py:129-142) and relays rank 0's JSON line -- the parent itself never touches a GPU.

A "step" is one pass of the hot path over one batch of synthetic input, inputs resident in HBM:
    pyramid tail: conv-BN-ReLU extras on the last backbone tap (SSD configs)        H2
    -> multi-scale heads forward (fused score|loc implicit GEMMs, all levels)       H1
    -> device-resident anchors (cached)                                             A1
    -> IoU match + target encode                                                    T1-T3
    -> hard-negative mining + multibox loss forward                                 S1, L1-L3
    -> loss backward + heads backward (dgrad, wgrad, dbias) + extras backward       L1, H1, H2
    -> (N > 1) RCCL all-reduce of the flat head-gradient bucket
    -> SGD step on the head parameters (stock torch optimizer: the runtime's, kept so no training work is skipped)
Workload: BASELINE.json configs[1] -- ssd_300_vgg16_voc, batch 32 per GPU, 81 classes (the literal of the sample file):
the two backbone taps (512@37x37, 512@18x18) are N(0,1) NHWC tensors (the backbone itself stays PyTorch-ROCm and is not part
of the path); the four extras blocks derive the remaining levels.  Weak scaling: per-GPU batch fixed.
The postprocess (eval) leg is timed separately and reported as nms_boxes_per_sec / postprocess_images_per_sec.
"""
import argparse
import gc as _pygc
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from single_shot_detection_amd import launch as _launch  # noqa: E402
from single_shot_detection_amd import synthetic as syn  # noqa: E402
from single_shot_detection_amd.distributed import GradBucket  # noqa: E402

# extras of the SSD sample files (samples/ssd_300_vgg16_voc.py:16-18, ssd_512_vgg16_coco.py)
TOWER = {'retina_rn50_500_coco': dict(num_layers=4, num_channels=256, kernel_size=3)}
# M2Det: the MLFPN neck (8 TUMs x 6 scales + SFAM, samples/m2det_512_vgg16_coco.py:10-17) sits between the two VGG taps and the heads
NECK = {'m2det_512_vgg16_coco': dict(taps=[(512, 64), (1024, 32)], num_scales=6, num_tums=8, base_reduced_channels=[512, 256])}
def _sgd(params, **kw):
    """torch.optim.SGD, single-pass (fused) implementation where this torch build has it (the default foreach form is four
    passes over parameters, gradients and momentum buffers)."""
    try:
        return torch.optim.SGD(params, fused=True, **kw)
    except (TypeError, RuntimeError, ValueError):
        return torch.optim.SGD(params, **kw)


EXTRAS = {'ssd_300_vgg16_voc': (('s', 512), ('s', 256), ('s', 256), ('s', 256)),
          'ssd_300_vgg16_voc_c21': (('s', 512), ('s', 256), ('s', 256), ('s', 256)),
          'ssd_512_vgg16_coco': (('s', 512), ('s', 256), ('s', 256), ('s', 256), ('s', 256))}

PEAK_FP32_MATRIX_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec
ACHIEVABLE_HBM_GBS = 6300.0       # ... and what a streaming kernel reaches on it (same section)


def measured_traffic(kernel_key):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/r05_pmc.json, else an earlier
    round's: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of this command by tools/r05_measure.sh; FETCH_SIZE doubled per
    the gfx950 correction in MI355X_MICROARCH.md, both in KiB).  None when no file holds this workload's kernel."""
    for name in ('r05_pmc.json', 'r04_pmc.json', 'r03_pmc.json'):
        try:
            d = json.load(open(os.path.join(REPO, 'profiles', name)))[kernel_key]
            return (2.0 * d['FETCH_SIZE_KiB'] + d['WRITE_SIZE_KiB']) * 1024.0
        except Exception:
            continue
    return None


def head_flops_per_image(levels, C):
    """sum over levels of 2*H*W*9*Cin*nb*(C+4)."""
    return sum(2.0 * h * h * 9 * cin * nb * (C + 4) for cin, h, nb in levels)


def head_gemm_time(hp):
    """(ms per step, algorithmic FLOPs per image) of the head GEMM launches the heads module timed inside the steps (heads.launch_events).
    One grouped launch per step, or -- with the dependency split (HotPath.overlap) -- the DOMINANT one: the levels taken from the
    backbone, on the main stream; the tail levels' small launch runs beside it on the second stream and is not added."""
    per_launch = {}
    for a, b, lv in hp.fwd_events:
        per_launch.setdefault(lv, []).append(a.elapsed_time(b))
    if not per_launch:
        return float('nan'), 0.0
    flops = {lv: sum(2.0 * h * w * 9 * cin * n for h, w, cin, n in lv) for lv in per_launch}
    if hp.overlap:
        lv = max(flops, key=flops.get)
        return float(np.sum(per_launch[lv])) / max(hp.fwd_steps, 1), flops[lv]
    return float(np.sum([t for ts in per_launch.values() for t in ts])) / max(hp.fwd_steps, 1), sum(flops.values())


class HotPath(object):
    def __init__(self, cfg_name, batch, device, seed=23):
        from single_shot_detection_amd.detection import detector_builder, anchor_generators, sampler
        from single_shot_detection_amd.detection.box_coder import BoxCoder
        from single_shot_detection_amd.detection.losses.multibox_loss import MultiboxLoss
        from single_shot_detection_amd.detection.postprocessor import Postprocessor
        from single_shot_detection_amd.detection.target_assigner import TargetAssigner
        import functools
        # the pyramid tail's weight gradients in one grouped launch at the end of the backward pass: scoped to train_step (the switch is
        # process-wide and changes what torch.autograd.grad returns for a convolution weight)
        self.defer_weight_gradients = True
        self.cfg = cfg = syn.CONFIGS[cfg_name]
        self.batch, self.device = batch, device
        self.levels, self.C = cfg['levels'], cfg['num_classes']
        torch.manual_seed(seed)
        self.heads = detector_builder.get_heads([l[0] for l in self.levels], [l[2] for l in self.levels], self.C,
                                                score_head_bias_init=(-4.6 if cfg['loss'] != 'ce_hnm' else 0.0)).to(device)
        # SSD configs: the backbone taps are the inputs, the extras (H2) derive the other levels; otherwise all levels are inputs
        self.extras = None
        n_in = len(self.levels)
        if cfg_name in EXTRAS:
            n_in = len(self.levels) - len(EXTRAS[cfg_name])
            self.extras = detector_builder.get_extras([self.levels[n_in - 1][0]], layers=EXTRAS[cfg_name]).to(device)
        # RetinaNet: the shared-conv tower (H3) sits between the FPN maps and the heads (retina_rn50_500_coco.py:18-24)
        self.tower = None
        if cfg_name in TOWER:
            from single_shot_detection_amd.detection.modules.predictors import SharedConvPredictor
            self.tower = SharedConvPredictor([l[0] for l in self.levels], [l[2] for l in self.levels], self.C, False, **TOWER[cfg_name]).to(device)
        self.neck = None
        in_levels = self.levels[:n_in]
        if cfg_name in NECK and not os.environ.get('SSDK_BENCH_NO_NECK'):
            from single_shot_detection_amd.bf.modules.features import MultilevelFeaturePyramid
            nk = dict(NECK[cfg_name])
            taps = nk.pop('taps')

            class _Taps(torch.nn.Module):   # only the channel counts of the two taps are needed to size the reducers
                def __init__(self):
                    super().__init__()
                    self.features = torch.nn.Sequential(torch.nn.Conv2d(3, taps[0][0], 1), torch.nn.Conv2d(taps[0][0], taps[1][0], 1))
            self.neck = MultilevelFeaturePyramid(_Taps(), out_layers=(0, 1), **nk).to(device)
            in_levels = [(c, h, 0) for c, h in taps]
        fm = syn.make_feature_maps(batch, in_levels, seed=seed)
        self.inputs = [torch.from_numpy(x).to(device).contiguous(memory_format=torch.channels_last).requires_grad_(True) for x in fm]
        p = dict(cfg['anchor'])
        gens = getattr(anchor_generators, p.pop('type')).build_anchor_generators(**p)
        img = torch.empty((1, 3, cfg['size'], cfg['size']), device=device)
        self.anchors = torch.cat([g.generate(img, (h, h)).reshape(-1) for g, (_, h, _) in zip(gens, self.levels)]).view(-1, 4)
        softmax = cfg['score_converter'] == 'SOFTMAX'
        gt = syn.make_ground_truth(batch, cfg['size'], self.C, seed=1, background=softmax)
        self.gt_np = gt
        self.gt = [torch.from_numpy(g).to(device) for g in gt]
        box_coder = BoxCoder(10.0, 5.0)
        if cfg['loss'] == 'ce_hnm':
            smp = functools.partial(sampler.hard_negative_mining, negative_per_positive_ratio=3, min_negative_per_image=5)
            cl = {'name': 'CrossEntropyLoss'}
        else:
            smp = sampler.naive_sampler
            cl = {'name': 'SigmoidFocalLoss', 'gamma': 2.0, 'alpha': 0.25}
        self.criterion = MultiboxLoss(sampler=smp, box_coder=box_coder, classification_loss=cl,
                                      localization_loss={'name': 'SmoothL1Loss'})
        self.assigner = TargetAssigner(cfg['matched'], cfg['unmatched'])
        self.post = Postprocessor(box_coder, score_threshold=0.01, nms={'max_per_class': 100, 'overlap_threshold': cfg['nms_thr']},
                                  score_converter=cfg['score_converter'], max_total=200)
        self.params = [p for p in self.heads.parameters()] + ([p for p in self.extras.parameters()] if self.extras is not None else []) + \
                      ([p for p in self.tower.parameters()] if self.tower is not None else []) + \
                      ([p for n, p in self.neck.named_parameters() if not n.startswith('base.')] if self.neck is not None else [])
        self.opt = _sgd(self.params, lr=1e-3, momentum=0.9, weight_decay=5e-4)
        self._one = torch.ones((), dtype=torch.float32, device=device)
        self.exchange = None    # N > 1: distributed.BucketedDataParallel around the path's modules (enable_exchange)
        # exchange step (N > 1): the head gradients are complete as soon as the heads' backward has run, so their ring starts
        # there and overlaps with the backward of the extras / tower; a second, small bucket carries the rest
        self.head_params = [p for p in self.heads.parameters()]
        self.rest_params = [p for p in self.params if all(p is not q for q in self.head_params)]
        # flat fp32 buckets; attach_() makes the weight-gradient kernels write straight into them (no pack / unpack pass per step)
        self.bucket_heads = GradBucket(self.head_params).attach_(device)
        self.bucket_rest = GradBucket(self.rest_params).attach_(device) if self.rest_params else None
        self.fwd_events = []
        self.fwd_steps = 0
        # dependency split (SSD configs): the heads of the backbone taps on the current stream, the pyramid tail and its levels' heads on a
        # second one -- forward and, through autograd's per-node streams, backward (detection/modules/heads.py multi_level_heads_split)
        # OFF by default: measured on MI355X (tools/overlap_probe.py, profiles/r04_overlap_*): replayed from a HIP graph the split step is
        # 1.5 % faster (3.20 -> 3.15 ms), enqueued eagerly it is 2 % slower (more autograd nodes, events and a second queue on the host)
        self.overlap = self.extras is not None and self.tower is None and self.neck is None and bool(os.environ.get('SSDK_OVERLAP'))
        self.side = None
        self.main_workgroups = int(os.environ.get('SSDK_MAIN_WGS', '0'))
        self.side_workgroups = int(os.environ.get('SSDK_SIDE_WGS', '0'))
        self.one_launch = not os.environ.get('SSDK_SPLIT_FORWARD')        # forward GEMM: all levels in one grouped launch (only the backward is split)
        self.ordered_backward = bool(os.environ.get('SSDK_ORDERED_BACKWARD'))

    def pyramid(self):
        sources = list(self.inputs)
        if self.neck is not None:
            sources, _ = self.neck.neck(sources)
        if self.extras is not None:
            x = sources[-1]
            if self.extras.training:
                from single_shot_detection_amd import ops
                ops.prepare_weight_transposes(self.extras)   # (what detection/detector.py does in front of the tail)
            for layer in self.extras:   # detector.py:39-43
                x = layer(x)
                sources.append(x)
        assert [tuple(s.shape[1:3]) for s in sources] == [(c, h) for c, h, _ in self.levels], [tuple(s.shape) for s in sources]
        return sources

    def forward_heads(self, timed=False):
        from single_shot_detection_amd.detection.modules.heads import multi_level_heads
        if self.overlap:
            return self.forward_heads_split(timed)
        sources = self.pyramid()
        score_sources = loc_sources = sources
        if self.tower is not None:
            score_sources, loc_sources = self.tower(sources)
        # timed: event pairs recorded by the heads module right around its library call(s) -- one grouped launch, or two when the score and
        # the loc tower feed separate maps (RetinaNet); self.fwd_steps counts the steps they belong to
        from single_shot_detection_amd.detection.modules import heads as heads_mod
        heads_mod.launch_events = self.fwd_events if timed else None
        try:
            out = multi_level_heads(score_sources, loc_sources, self.heads)
        finally:
            heads_mod.launch_events = None
        self.fwd_steps += 1 if timed else 0
        return out

    def forward_heads_split(self, timed=False):
        from single_shot_detection_amd import ops
        from single_shot_detection_amd.detection.modules import heads as heads_mod
        if self.side is None and not os.environ.get('SSDK_OVERLAP_ONE_STREAM'):
            # (default priority: a HIGH-priority queue is served strictly first on this hardware -- the main GEMM's dispatch packet was not
            # looked at until the whole tail chain had drained, tools/graph_branch_probe.py)
            self.side = torch.cuda.Stream(priority=int(os.environ.get('SSDK_SIDE_PRIORITY', '0')))

        def run_tail():
            x, outs = self.inputs[-1], []
            if self.extras.training:
                ops.prepare_weight_transposes(self.extras)
            for layer in self.extras:
                x = layer(x)
                outs.append(x)
            return outs
        heads_mod.launch_events = self.fwd_events if timed else None
        try:
            scores, locs, sources = heads_mod.multi_level_heads_split(self.inputs, self.heads, len(self.inputs), run_tail, side_stream=self.side,
                                                                       main_workgroups=self.main_workgroups, side_workgroups=self.side_workgroups,
                                                                       one_launch=self.one_launch, ordered_backward=self.ordered_backward)
        finally:
            heads_mod.launch_events = None
        self.fwd_steps += 1 if timed else 0
        assert [tuple(s.shape[1:3]) for s in sources] == [(c, h) for c, h, _ in self.levels], [tuple(s.shape) for s in sources]
        return scores, locs

    def train_step(self, world=1, timed=False):
        from single_shot_detection_amd import ops
        with ops.deferred_weight_gradients(self.defer_weight_gradients):
            return self._train_step(world, timed)

    def enable_exchange(self, process_group=None):
        """N > 1: the path's modules inside distributed.BucketedDataParallel -- the SAME wrapper detection.init(distributed=True) puts
        around the predictor (the role of apex DDP, detection/init.py:80-86): rank 0's parameters broadcast, two flat fp32 buckets
        (heads / the rest), the heads' ring started by the hook of the last head parameter, i.e. under the backward pass of the
        pyramid tail, both rings awaited at the end of backward()."""
        from single_shot_detection_amd.distributed import BucketedDataParallel
        if self.exchange is None:
            mods = torch.nn.ModuleDict({k: m for k, m in (('heads', self.heads), ('extras', self.extras), ('tower', self.tower), ('neck', self.neck)) if m is not None})
            for n, p in mods.named_parameters():   # (the backbone stand-in inside the M2Det neck is not part of the path)
                if all(p is not q for q in self.params):
                    p.requires_grad_(False)
            groups = [g for g in (self.head_params, self.rest_params) if g]
            self.exchange = BucketedDataParallel(mods, groups=groups, process_group=process_group)
            self.exchange.collect_timing = True   # three events per bucket and step: the line says how long each ring took and how long its join stalled
            self.bucket_heads = self.exchange.buckets[0]
            self.bucket_rest = self.exchange.buckets[1] if len(self.exchange.buckets) > 1 else None
        return self.exchange

    def _train_step(self, world=1, timed=False):
        if world > 1 and self.exchange is None:
            self.enable_exchange()
        self.opt.zero_grad(set_to_none=True)
        for s in self.inputs:
            s.grad = None
        scores, locs = self.forward_heads(timed)
        target = self.assigner.encode_ground_truth(self.resident_ground_truth(), self.anchors)
        loss, class_loss, loc_loss = self.criterion((scores, locs), self.anchors, target)
        # (the root gradient is a resident 1.0: backward() would make a ones_like(loss) with a fill launch every step; N > 1: the exchange
        # runs inside -- BucketedDataParallel's hooks and end-of-backward callback)
        loss.backward(self._one)
        self.opt.step()
        return loss

    def resident_ground_truth(self):
        """The step's ground truth in the library's packed device layout (target_assigner.PackedGroundTruth: rows + offsets), made ONCE
        for the synthetic batch: the inputs are resident in HBM when the timed region starts -- a list of per-image device tensors would
        be concatenated and its offsets copied in every step (two launches that belong to the data loader, not to the path)."""
        from single_shot_detection_amd.detection.target_assigner import PackedGroundTruth
        if isinstance(self.gt, PackedGroundTruth):
            return self.gt
        if getattr(self, '_packed_for', None) is not self.gt:
            self._packed = PackedGroundTruth.from_list(self.gt, self.device)
            self._packed_for = self.gt
        return self._packed

    def set_training(self, training):
        """model.train() / model.eval() of the path's modules (the reference evaluates under model.eval(): BatchNorm on running statistics)."""
        for m in (self.extras, self.tower, self.neck, self.heads):
            if m is not None:
                m.train(training)

    def eval_step(self):
        with torch.no_grad():
            scores, locs = self.forward_heads()
            return self.post.postprocess_padded((scores, locs), self.anchors)


def cpu_baseline(hp, probe_images=4):
    """The oracle (CPU restatement, kind 'port') timed on the host cores on a bounded sample of the same workload: the C oracle (OpenMP)
    for match / HNM / loss fwd+bwd over the full batch, torch CPU convolutions for the heads fwd+bwd over the FULL batch at the best of
    {16, 32, 64, 128} threads (picked on a `probe_images` sample: an oversubscribed pool made the round-3 figure wander 5..10 images/s
    between boxes); reported per image.  Calibration against the reference itself: BASELINE.md section 4 (the port is 1.2x .. 18x
    FASTER than the reference's Python on the match / loss stages and identical -- torch CPU convolutions -- on the conv leg)."""
    import oracle
    import torch.nn.functional as F
    cfg, C, B = hp.cfg, hp.C, hp.batch
    A = hp.anchors.shape[0]
    anchors = hp.anchors.cpu().numpy()
    logits = syn.make_logits(B, A, C, seed=2)
    locs = syn.make_locs(B, A, seed=3, scale=0.5)

    def small():
        target = oracle.encode_ground_truth(hp.gt_np, anchors, cfg['matched'], cfg['unmatched'])
        if cfg['loss'] == 'ce_hnm':
            mask = oracle.hard_negative_mining(logits, target, 3, 5)
            oracle.multibox_loss(logits, locs, anchors, target, mask, kind='ce')
        else:
            mask = oracle.naive_sampler(logits, target)
            oracle.multibox_loss(logits, locs, anchors, target, mask, kind='focal')
    cores = os.cpu_count() or 1
    # the oracle's OpenMP loops at the best of {16, 32, 64, 128} threads (one timed pass each after a warm-up pass; round 4 ran them at
    # omp_get_max_threads() = every core of the box and measured 0.38 .. 6.3 ms / image depending on the box), then the best of three there
    small()
    oracle_probe = {}
    for t in sorted({max(1, min(t, cores)) for t in (16, 32, 64, 128)}):
        oracle.set_threads(t)
        small()
        t0 = time.perf_counter()
        small()
        oracle_probe[t] = (time.perf_counter() - t0) / B
    omp_threads = min(oracle_probe, key=oracle_probe.get)
    oracle.set_threads(omp_threads)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        small()
        ts.append(time.perf_counter() - t0)
    t_small = min(ts) / B
    with torch.no_grad():
        srcs = hp.pyramid()
    ws = [(h['score'].weight.detach().cpu().contiguous().requires_grad_(True), h['score'].bias.detach().cpu().requires_grad_(True),
           h['loc'].weight.detach().cpu().contiguous().requires_grad_(True), h['loc'].bias.detach().cpu().requires_grad_(True)) for h in hp.heads]

    def conv_leg(n_img):
        xs = [s.detach()[:n_img].cpu().contiguous().requires_grad_(True) for s in srcs]
        t0 = time.perf_counter()
        outs = []
        for x, (w1, b1, w2, b2) in zip(xs, ws):
            outs.append(F.conv2d(x, w1, b1, padding=1).permute(0, 2, 3, 1).reshape(n_img, -1))
            outs.append(F.conv2d(x, w2, b2, padding=1).permute(0, 2, 3, 1).reshape(n_img, -1))
        torch.cat(outs, 1).sum().backward()
        return (time.perf_counter() - t0) / n_img
    prev_threads = torch.get_num_threads()
    probe = {}
    sb = min(probe_images, B)
    for t in sorted({min(t, cores) for t in (16, 32, 64, 128)}):
        torch.set_num_threads(t)
        conv_leg(1)                       # (first call at a thread count: pool start-up)
        probe[t] = conv_leg(sb)
    best_t = min(probe, key=probe.get)
    torch.set_num_threads(best_t)
    t_conv = conv_leg(B)
    torch.set_num_threads(prev_threads)
    per_image = t_small + t_conv
    return {'value': 1.0 / per_image, 'unit': 'images/sec', 'cores': max(best_t, omp_threads), 'kind': 'port',
            'host_cores': cores, 'conv_threads': best_t, 'oracle_threads': omp_threads,
            'probe_ms_per_image_by_threads': {str(k): v * 1e3 for k, v in probe.items()},
            'oracle_probe_ms_per_image_by_threads': {str(k): v * 1e3 for k, v in oracle_probe.items()},
            'sample': f'oracle match+HNM+loss fwd/bwd on {B} images, best of 3 ({t_small * 1e3:.2f} ms/img at {omp_threads} OpenMP threads, the best of {sorted(oracle_probe)}) + torch CPU head convs '
                      f'fwd+bwd on the full batch of {B} images at {best_t} threads, the best of {sorted(probe)} on a {sb}-image probe '
                      f'({t_conv * 1e3:.1f} ms/img)'}


def quiesce():
    """Collect Python garbage NOW: main() switches the automatic collector off, so that a full collection (~100 ms with the synthetic inputs
    and modules of a few configurations alive: one eager step of fast_mode_legs measured 109 ms, host and device, in the middle of ten
    2.4 ms ones) never lands inside a timed region; every leg calls this before it starts timing."""
    _pygc.collect()


def gpu_time_us(fn, inner=10, reps=5):
    """Median device time of one ``fn()`` in microseconds: ``inner`` back-to-back calls between two events on the launch stream, behind
    ~1 ms of queued spin so that the host is ahead of the GPU when the first one starts (a 20 us call is otherwise timed as the
    host's enqueue rate)."""
    fn()
    torch.cuda.synchronize()
    quiesce()
    out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(2_000_000)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / inner)
    return float(np.median(out))


def hbm_legs(device, cfg_name='ssd_300_vgg16_voc', batch=64):
    """north_star's batch-64 HBM check: the streaming calls of the path on device-resident synthetic inputs, each
    event-timed on its own; achieved = algorithmic bytes / time, against the 8 TB/s HBM3E figure."""
    import ctypes
    from single_shot_detection_amd import _lib
    hp = HotPath(cfg_name, batch, device)
    A, C, B = hp.anchors.shape[0], hp.C, batch
    legs = {}

    # what a kernel boundary costs on this box: dependent one-element launches back to back (device time per launch)
    one = torch.zeros((1,), device=device)

    def chain():
        for _ in range(20):
            one.add_(1.0)
    boundary_us = gpu_time_us(chain, inner=5) / 20.0

    def leg(name, us, nbytes, note, launches):
        # floor_us: the larger of (algorithmic bytes at the 6.3 TB/s a streaming kernel reaches on this part, MI355X_MICROARCH.md) and
        # (the call's dependent launches x the measured boundary): says whether a leg is bandwidth-short or launch-bound
        gbs = nbytes / (us * 1e-6) / 1e9
        bw_floor = nbytes / ACHIEVABLE_HBM_GBS / 1e9 * 1e6
        legs[name] = {'us': us, 'algorithmic_bytes': nbytes, 'achieved': gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': gbs / PEAK_HBM_GBS,
                      'launches': launches, 'floor_us': max(bw_floor, launches * boundary_us), 'bandwidth_floor_us': bw_floor,
                      'launch_floor_us': launches * boundary_us, 'bound': 'launch' if launches * boundary_us > bw_floor else 'hbm',
                      'of_floor': max(bw_floor, launches * boundary_us) / us, 'what': note}

    # T1-T3: IoU match + target encode, 41*A bytes per image (anchors 16*A read once per image + 24*A target + mask); ground truth packed once
    # (the packing is host work that runs ahead of the stream), the library call itself is what is timed
    from single_shot_detection_amd.detection.target_assigner import pack_ground_truth
    lib = _lib.lib()
    rows, offs, total = pack_ground_truth(hp.gt, device)
    tgt = torch.empty((B, A, 6), dtype=torch.float32, device=device)
    ews = torch.empty((max(lib.ssdk_encode_ground_truth_workspace_bytes(B, total), 4096),), dtype=torch.uint8, device=device)
    leg('encode_ground_truth', gpu_time_us(lambda: _lib.check(lib.ssdk_encode_ground_truth(
        _lib.ptr(rows), 6, _lib.ptr(offs), B, total, _lib.ptr(hp.anchors), A, 0.5, 0.5, _lib.ptr(tgt), None, _lib.ptr(ews), ews.numel(),
        _lib.current_stream()), 'encode')), 41.0 * A * B, 'ssdk_encode_ground_truth (IoU + matcher + target rows): gt_argmax_kernel + assign_kernel', 2)
    logits = torch.from_numpy(syn.make_logits(B, A, C, seed=2)).to(device)
    locs = torch.from_numpy(syn.make_locs(B, A, seed=3, scale=0.5)).to(device)
    trained = logits.clone().view(B, A, C)
    if hp.cfg['score_converter'] == 'SOFTMAX':
        trained[..., 0] += 6.0      # background logit + 6: ~5 % of the (anchor, class) pairs pass the threshold
    else:
        trained -= 6.25             # sigmoid scores: the same share of pairs passes (logit(0.01) = -4.6 is 1.65 sigma above the mean)
    trained = trained.view(B, -1)
    # launches: sample plan (long lists) select x 2 + class bound, short lists select only; then the NMS head and the per-image finish
    # (image bound + the few classes to redo + merge: post_finish_kernel, round 5), or -- where the two-pass NMS does not apply
    # (postprocess.hip nms_head_size) -- one NMS launch and the merge
    for tag, sc, launches in (('worst_case', logits, 5), ('trained_like', trained, 3)):
        us = gpu_time_us(lambda: hp.post.postprocess_padded((sc, locs), hp.anchors), inner=5)
        leg(f'postprocess_{tag}', us, 4.0 * A * (C + 4) * B, 'ssdk_postprocess: score convert + threshold + per-class top-100 + decode + NMS + top-200', launches)
        legs[f'postprocess_{tag}']['nms_candidates_per_image'] = float(hp.post.last_nms_candidates.sum().item()) / B
    # the same trained-like statistics, but a DIFFERENT batch in every call (two batches taking turns): the per-class hot bound the
    # postprocess carries from one call to the next (postprocess.hip HotState) then comes from other images than the ones it is used on
    other = torch.from_numpy(syn.make_logits(B, A, C, seed=12)).to(device).view(B, A, C)
    if hp.cfg['score_converter'] == 'SOFTMAX':
        other[..., 0] += 6.0
    else:
        other -= 6.25
    other = other.view(B, -1)
    turn = [0]

    def alternating():
        turn[0] ^= 1
        return hp.post.postprocess_padded((other if turn[0] else trained, locs), hp.anchors)
    leg('postprocess_trained_like_alternating_batches', gpu_time_us(alternating, inner=6), 4.0 * A * (C + 4) * B,
        'ssdk_postprocess on two different trained-like batches taking turns (the bound of one call was left by the other batch)', 3)
    # S1 + L1: sampler (reads the logits once), loss forward, loss backward (writes dscores + dlocs)
    target = hp.assigner.encode_ground_truth(hp.gt, hp.anchors)
    from single_shot_detection_amd.detection import sampler as smp
    ws = smp.loss_workspace(B, A, C, device)
    mask = torch.empty((B, A), dtype=torch.uint8, device=device)
    cls_col = target[..., 4]
    if hp.cfg['loss'] == 'ce_hnm':
        leg('hard_negative_mining', gpu_time_us(lambda: _lib.check(lib.ssdk_hard_negative_mining(
            _lib.ptr(logits), cls_col.data_ptr(), 6, B, A, C, 3.0, 5, _lib.ptr(mask), _lib.ptr(ws), ws.numel(), _lib.current_stream()), 'hnm')),
            4.0 * A * C * B, 'ssdk_hard_negative_mining: hnm_rows (log-sum-exp of every row) + hnm_select', 2)
    else:
        mask.copy_((cls_col > 0).to(torch.uint8))   # naive_sampler: positives only (sampler.py:9-10)
    params = hp.criterion.loss_params()
    out3 = torch.empty((3,), dtype=torch.float32, device=device)
    tgt2 = target.clone()
    # (the call encodes the target's box columns in place; after the first call they hold encoded values and the next calls re-encode
    # those -- different numbers, the same reads, arithmetic and writes: what is timed is the kernel, not a copy that restores the target)
    leg('multibox_loss_fwd', gpu_time_us(lambda: _lib.check(lib.ssdk_multibox_loss_fwd(
        ctypes.byref(params), _lib.ptr(logits), _lib.ptr(locs), _lib.ptr(hp.anchors), _lib.ptr(tgt2), _lib.ptr(mask), B, A, C, 1, _lib.ptr(out3),
        _lib.ptr(ws), ws.numel(), _lib.current_stream()), 'loss_fwd')), (24.0 + 16.0 + 1.0 + 16.0) * A * B,
        'ssdk_multibox_loss_fwd: target rows + locs + mask in, encoded box columns out (classification term by gather on the sampled rows): loss_fwd_kernel + loss_finalize_kernel', 2)
    dsc, dlo = torch.empty_like(logits), torch.empty_like(locs)
    gout = torch.ones((2,), dtype=torch.float32, device=device)
    leg('multibox_loss_bwd', gpu_time_us(lambda: _lib.check(lib.ssdk_multibox_loss_bwd(
        ctypes.byref(params), _lib.ptr(logits), _lib.ptr(locs), _lib.ptr(hp.anchors), _lib.ptr(tgt2), _lib.ptr(mask), _lib.ptr(gout), B, A, C,
        _lib.ptr(dsc), _lib.ptr(dlo), _lib.ptr(ws), ws.numel(), _lib.current_stream()), 'loss_bwd')), (4.0 * C + 16.0) * A * B,
        'ssdk_multibox_loss_bwd: writes dscores + dlocs in full', 1)
    return {'workload': f'{cfg_name} batch {batch}, A={A}, C={C}', 'kernel_boundary_us': boundary_us, 'legs': legs}


PER_CONFIG = (('ssd_300_vgg16_voc', 64), ('ssd_300_vgg16_voc_c21', 32), ('ssd_mb2_voc', 2), ('ssd_512_vgg16_coco', 16),
              ('retina_rn50_500_coco', 32), ('m2det_512_vgg16_coco', 16))


def serving_legs(device, cfg_name='ssd_300_vgg16_voc', batches=(1, 2, 8), reps=30):
    """Evaluation step (pyramid tail + heads forward + postprocess) at serving batch sizes: enqueued launch by launch, and replayed from a
    HIP graph (single_shot_detection_amd/graphs.py).  Images per second of both."""
    from single_shot_detection_amd.graphs import GraphedCallable
    out = []
    for b in batches:
        hp = HotPath(cfg_name, b, device)
        hp.set_training(False)

        def step(*taps):
            hp.inputs = list(taps)
            return hp.eval_step()

        def rate(fn):
            for _ in range(3):
                fn()
            quiesce()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return b * reps / (time.perf_counter() - t0)

        taps = [t.detach().clone() for t in hp.inputs]
        eager = rate(lambda: step(*taps))
        graphed = GraphedCallable(step, taps)
        replay = rate(lambda: graphed(*taps))
        out.append({'config': cfg_name, 'batch': b, 'eager_images_per_sec': eager, 'graph_images_per_sec': replay})
        del hp, graphed
        torch.cuda.empty_cache()
    return out


def train_graph_legs(device, cases=(('ssd_300_vgg16_voc', 32), ('ssd_mb2_voc', 2), ('ssd_300_vgg16_voc', 2)), reps=30):
    """The WHOLE training step (forward, match, sampler, loss, backward, fused SGD) captured in a HIP graph and replayed, against the
    step enqueued launch by launch.  The ground truth sits in static device buffers (target_assigner.PackedGroundTruth); a training
    loop would refill them between replays.  ms per step of both."""
    from single_shot_detection_amd.detection.target_assigner import PackedGroundTruth
    from single_shot_detection_amd.graphs import GraphedCallable
    out = []
    for cfg_name, b in cases:
        hp = HotPath(cfg_name, b, device)
        hp.gt = PackedGroundTruth.from_list(hp.gt, device)

        def ms(fn):
            for _ in range(3):
                fn()
            quiesce()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e3

        eager = ms(hp.train_step)
        graphed = GraphedCallable(hp.train_step, [])
        replay = ms(graphed)
        out.append({'config': cfg_name, 'batch': b, 'eager_ms_per_step': eager, 'graph_ms_per_step': replay})
        del hp, graphed
        torch.cuda.empty_cache()
    return out


def step_fn_legs(device, cases=(('ssd_300_vgg16_voc_c21', 21, 32), ('ssd_300_vgg16_voc', 81, 2)), reps=20):
    """The product API itself: detection.init(...) -> step_fn(step, 'train', (imgs, ground truth), state), loss.backward(), SGD on the
    predictor's head-side parameters -- eager, and with graph_hot_path=True (the libssdk part of the step replayed from two HIP graphs
    behind the eager PyTorch backbone).  The stand-in VGG16-BN backbone is excluded from the hot-path figure by device events around the
    segment's two replays (graphs.GraphedSegment.collect_timing: input copies + forward graph; upstream gradient + backward graph +
    gradient hand-over); ``hot_ms`` = forward + backward + the fused SGD step on those parameters, to be read against
    ``graph_replay_ms_per_step`` of the same configuration (the whole bench step captured as one graph, per_config / train_graph)."""
    from single_shot_detection_amd.detection import init as det_init
    out = []
    for cfg_name, ncls, b in cases:
        cfg = syn.CONFIGS[cfg_name if cfg_name in syn.CONFIGS else 'ssd_300_vgg16_voc']
        model = {'base': {'name': 'torchvision_vgg16_bn', 'pretrained': False},
                 'detector': {'num_classes': ncls, 'use_depthwise': False,
                              'features': {'name': 'Features', 'out_layers': (32, 42), 'last_feature_layer': 42},
                              'extras': {'layers': (('s', 512), ('s', 256), ('s', 256), ('s', 256))}},
                 'anchor_generator': dict(cfg['anchor'])}
        args = ({'xy_scale': 10.0, 'wh_scale': 5.0},
                {'score_threshold': .01, 'max_total': 200, 'nms': {'max_per_class': 100, 'overlap_threshold': .45}, 'score_converter': 'SOFTMAX'},
                {'classification_loss': {'name': 'CrossEntropyLoss'}, 'localization_loss': {'name': 'SmoothL1Loss'},
                 'classification_weight': 1.0, 'localization_weight': 1.0},
                {'name': 'hard_negative_mining', 'negative_per_positive_ratio': 3, 'min_negative_per_image': 5},
                {'matched_threshold': 0.5, 'unmatched_threshold': 0.5})
        imgs = torch.randn((b, 3, cfg['size'], cfg['size']), device=device)
        gt = [torch.from_numpy(g) for g in syn.make_ground_truth(b, cfg['size'], ncls, seed=1)]
        res = {'config': cfg_name, 'batch': b}
        for graphed in (False, True):
            torch.manual_seed(0)
            wrapper, init_state, step_fn = det_init.init(device, model, *args, graph_hot_path=graphed)
            det = wrapper.model
            det.train()
            hot_params = [p for n, p in det.predictor.named_parameters() if not n.startswith('features.')]
            opt = torch.optim.SGD(hot_params, lr=1e-4, momentum=0.9, fused=True)
            state = init_state()

            def one(k):
                opt.zero_grad(set_to_none=True)
                det.zero_grad(set_to_none=True)
                loss, _, st = step_fn(k, 'train', (imgs, gt), state)
                loss.backward()
                opt.step()
            for k in range(4):
                one(k)
            quiesce()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(reps):
                one(k)
            torch.cuda.synchronize()
            res['graphed_step_fn_ms' if graphed else 'eager_step_fn_ms'] = (time.perf_counter() - t0) / reps * 1e3
            if graphed:
                seg = next(iter(step_fn.hot_segments.values())).segment
                seg.collect_timing = True
                fwd, bwd, sgd = [], [], []
                for k in range(reps):
                    opt.zero_grad(set_to_none=True)
                    det.zero_grad(set_to_none=True)
                    loss, _, st = step_fn(k, 'train', (imgs, gt), state)
                    loss.backward()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    opt.step()
                    e1.record()
                    f, w = seg.read_timing()
                    e1.synchronize()
                    fwd.append(f); bwd.append(w); sgd.append(e0.elapsed_time(e1))
                res['hot_forward_ms'], res['hot_backward_ms'], res['hot_sgd_ms'] = (float(np.median(v)) for v in (fwd, bwd, sgd))
                res['hot_ms'] = res['hot_forward_ms'] + res['hot_backward_ms'] + res['hot_sgd_ms']
            del wrapper, step_fn, det, opt
            torch.cuda.empty_cache()
        # the bench's own step of the same configuration, captured whole
        from single_shot_detection_amd.detection.target_assigner import PackedGroundTruth
        from single_shot_detection_amd.graphs import GraphedCallable
        hp = HotPath(cfg_name, b, device)
        hp.gt = PackedGroundTruth.from_list(hp.gt, device)
        g = GraphedCallable(hp.train_step, [])
        for _ in range(3):
            g()
        quiesce()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            g()
        torch.cuda.synchronize()
        res['graph_replay_ms_per_step'] = (time.perf_counter() - t0) / reps * 1e3
        res['hot_over_replay'] = res['hot_ms'] / res['graph_replay_ms_per_step']
        out.append(res)
        del hp, g
        torch.cuda.empty_cache()
    return out


PEAK_BF16_MATRIX_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA (the 5 PF headline figure includes 2:1 sparsity)


def fast_mode_legs(device, cfg_name='ssd_300_vgg16_voc', batch=32, steps=10):
    """The opt-in split-bf16 mode of the forward head GEMM (heads.set_fast_mode('bf16x3'); reference analogue: apex AMP O1,
    bf/training/env.py:87-95), reported on its own: never part of `value` or `roofline`.  Its time against the fp32 launch on the same
    inputs, its error against the fp32 outputs and loss, the train step with it switched on, and its fraction of the BF16 peak --
    algorithmic FLOPs (what the fp32 kernel is priced with) and issued FLOPs (three bf16 products per fp32 product)."""
    from single_shot_detection_amd.detection.modules import heads as heads_mod
    from single_shot_detection_amd.detection.modules.heads import multi_level_heads
    hp = HotPath(cfg_name, batch, device)
    for _ in range(2):
        hp.train_step()
    with torch.no_grad():
        srcs = [t.detach() for t in hp.pyramid()]
        us32 = gpu_time_us(lambda: multi_level_heads(srcs, srcs, hp.heads), inner=5)
        s32, l32 = multi_level_heads(srcs, srcs, hp.heads)
        with heads_mod.fast_mode('bf16x3'):
            usf = gpu_time_us(lambda: multi_level_heads(srcs, srcs, hp.heads), inner=5)
            sf, lf = multi_level_heads(srcs, srcs, hp.heads)
    err_s = float((sf - s32).abs().max()) / float(s32.abs().max())
    err_l = float((lf - l32).abs().max()) / float(l32.abs().max())
    losses = []
    for sc, lo in ((s32, l32), (sf, lf)):
        target = hp.assigner.encode_ground_truth(hp.gt, hp.anchors)
        losses.append(float(hp.criterion((sc, lo), hp.anchors, target)[0]))
    with heads_mod.fast_mode('bf16x3'):
        for _ in range(2):
            hp.train_step()
        quiesce()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            hp.train_step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
    flops = head_flops_per_image(hp.levels, hp.C) * batch
    tf = flops / (usf * 1e-6) / 1e12
    del hp
    torch.cuda.empty_cache()
    tower = fast_mode_tower_leg(device)
    return {'mode': 'bf16x3', 'tower': tower, 'what': 'forward head GEMM with operands split into bf16 pieces, a_hi b_hi + a_hi b_mid + a_mid b_hi on '
                                      'v_mfma_f32_32x32x16_bf16, fp32 accumulate; weights split per call (included in the time); in the training step the dense data gradients and the weight gradients take the same mode',
            'workload': f'{cfg_name} batch {batch}', 'heads_fwd_us': usf, 'heads_fwd_fp32_us': us32, 'speedup_vs_fp32_launch': us32 / usf,
            'max_abs_err_over_scale': {'scores': err_s, 'locs': err_l}, 'loss_fp32': losses[0], 'loss_fast': losses[1],
            'loss_abs_diff': abs(losses[0] - losses[1]),
            'train_step_ms': ms, 'train_images_per_sec': batch / (ms * 1e-3),
            'roofline': {'bound': 'mfma', 'dtype': 'bf16', 'peak': PEAK_BF16_MATRIX_TFLOPS, 'unit': 'TFLOP/s', 'achieved_algorithmic': tf,
                         'frac_algorithmic': tf / PEAK_BF16_MATRIX_TFLOPS, 'achieved_issued': 3.0 * tf, 'frac_issued': 3.0 * tf / PEAK_BF16_MATRIX_TFLOPS}}


def fast_mode_tower_leg(device, cfg_name='retina_rn50_500_coco', batch=32):
    """The same mode on the generic convolutions (ssdk_conv2d_fwd_fast): RetinaNet's two 4-layer towers + heads, forward (evaluation mode:
    what a serving step runs), fp32 against split-bf16 on the same inputs."""
    from single_shot_detection_amd.detection.modules import heads as heads_mod
    from single_shot_detection_amd.detection.modules.heads import multi_level_heads
    hp = HotPath(cfg_name, batch, device)
    hp.set_training(False)

    def fwd():
        with torch.no_grad():
            srcs = hp.pyramid()
            ssrc, lsrc = (srcs, srcs) if hp.tower is None else hp.tower(srcs)
            return multi_level_heads(ssrc, lsrc, hp.heads)
    us32 = gpu_time_us(fwd, inner=3, reps=3)
    s32, l32 = fwd()
    with heads_mod.fast_mode('bf16x3'):
        usf = gpu_time_us(fwd, inner=3, reps=3)
        sf, lf = fwd()
    # the whole training step: fp32 against forward + data gradients + weight gradients in the split-bf16 mode
    hp.set_training(True)

    def step_ms(n=8):
        for _ in range(3):
            hp.train_step()
        quiesce()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            hp.train_step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    ms32 = step_ms()
    with heads_mod.fast_mode('bf16x3'):
        msf = step_ms()
    return {'workload': f'{cfg_name} batch {batch}: towers + heads forward, evaluation mode', 'fwd_us': usf, 'fwd_fp32_us': us32,
            'speedup_vs_fp32': us32 / usf,
            'max_abs_err_over_scale': {'scores': float((sf - s32).abs().max()) / float(s32.abs().max()),
                                       'locs': float((lf - l32).abs().max()) / float(l32.abs().max())},
            'train_step': {'what': 'full training step (towers + heads fwd / bwd, focal loss, SGD): fp32 against forward + data gradients in the '
                                   'split-bf16 mode, weight gradients in it too (igemm_wgrad_bf16x3_kernel)', 'fp32_ms': ms32, 'fast_ms': msf, 'speedup': ms32 / msf,
                           'images_per_sec_fast': batch / (msf * 1e-3)}}


def graph_replay_leg(hp, device, n):
    """The same training step (same kernels, same arguments) captured once in a HIP graph and replayed (single_shot_detection_amd/graphs.py):
    `ms_per_step` above is the step as the reference's loop would drive it, launch by launch from Python -- for the configs with hundreds of
    small launches per step (M2Det's neck, RetinaNet's towers, MobileNet's tail at batch 2) that is the host's enqueue rate, not the GPU.
    Reported beside it, never instead of it."""
    from single_shot_detection_amd.detection.target_assigner import PackedGroundTruth
    from single_shot_detection_amd.graphs import GraphedCallable
    try:
        hp.fwd_events = []
        hp.fwd_steps = 0
        hp.gt = PackedGroundTruth.from_list(hp.gt, device, capacity=sum(len(g) for g in hp.gt) + 7)
        step = GraphedCallable(hp.train_step, [], warmup=2)
        step()
        quiesce()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        loss = float(step.static_out.detach())
        if not np.isfinite(loss):
            raise RuntimeError('loss after the replays is %r' % loss)
        return {'graph_replay_ms_per_step': dt * 1e3, 'graph_replay_images_per_sec': hp.batch / dt}
    except Exception as e:   # (reported, not fatal: the eager number above is the leg's result)
        return {'graph_replay_ms_per_step': None, 'graph_replay_error': '%s: %s' % (type(e).__name__, e)}


def deterministic_leg(hp, n):
    """The same training step under ops.set_deterministic (the reference's cudnn.deterministic = True, bf/training/env.py:74-76): no fp32
    atomics -- dense data / weight gradients for the heads, no K splits, ordered reductions.  ms per step, reported beside the default."""
    from single_shot_detection_amd import ops
    try:
        with ops.deterministic():
            for _ in range(2):
                hp.train_step()
            quiesce()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                hp.train_step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
        return {'deterministic_ms_per_step': dt * 1e3, 'deterministic_images_per_sec': hp.batch / dt}
    except Exception as e:   # (reported, not fatal)
        return {'deterministic_ms_per_step': None, 'deterministic_error': '%s: %s' % (type(e).__name__, e)}


def per_config_legs(device, steps=4, warmup=2):
    """The other BASELINE.json configs (parity-test cases, not the headline): a few train steps each."""
    out = []
    for name, batch in PER_CONFIG:
        import gc
        hp = HotPath(name, batch, device)
        for _ in range(warmup):
            hp.train_step()
        quiesce()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hp.train_step()
        torch.cuda.synchronize()
        # a host-bound config (ssd_mb2_voc at batch 2: < 1 ms per step) is timed over more steps, with the collector out of the way: one
        # generation-2 collection of the previous configs' garbage inside four timed steps read as 22 ms per step in one pass of round 3
        n = steps if time.perf_counter() - t0 > 5e-3 else 10 * steps
        gc.collect()
        gc_was_on = gc.isenabled()   # (main() already runs with the collector off; a caller that imported this module may not)
        gc.disable()
        try:
            quiesce()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                hp.train_step(timed=True)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
        finally:
            if gc_was_on:
                gc.enable()
        fwd_ms, fl_img = head_gemm_time(hp)
        tf = fl_img * batch / (fwd_ms * 1e-3) / 1e12
        sc, lo = hp.forward_heads()
        sc, lo = sc.detach(), lo.detach()
        us = gpu_time_us(lambda: hp.post.postprocess_padded((sc, lo), hp.anchors), inner=3, reps=3)
        # the head GEMM launch alone, back to back with the host ahead of the GPU: `head_gemm_ms` above is the interval between two events
        # INSIDE the training step, which for a step the host can barely keep ahead of (the 21-class SSD-300 at 2.4 ms, MobileNet at
        # batch 2) also holds the host's time to prepare the call
        from single_shot_detection_amd.detection.modules.heads import multi_level_heads
        with torch.no_grad():
            srcs = hp.pyramid()
            ssrc, lsrc = (srcs, srcs) if hp.tower is None else hp.tower(srcs)
            alone_ms = gpu_time_us(lambda: multi_level_heads(ssrc, lsrc, hp.heads), inner=5, reps=5) * 1e-3
        tf_alone = head_flops_per_image(hp.levels, hp.C) * batch / (alone_ms * 1e-3) / 1e12
        row = {'config': name, 'per_gpu_batch': batch, 'ms_per_step': dt * 1e3, 'images_per_sec': batch / dt, 'head_gemm_ms': fwd_ms,
               'head_gemm_tflops': tf, 'head_gemm_frac': tf / PEAK_FP32_MATRIX_TFLOPS,
               'head_gemm_alone_ms': alone_ms, 'head_gemm_alone_frac': tf_alone / PEAK_FP32_MATRIX_TFLOPS,
               'postprocess_worst_case_images_per_sec': batch / (us * 1e-6)}
        del srcs, ssrc, lsrc
        del sc, lo
        row.update(deterministic_leg(hp, n))
        row.update(graph_replay_leg(hp, device, n))
        out.append(row)
        del hp
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)    # (50 x 3 ms: a timed region of 0.15 s; 20 steps after 3 warm-up steps read 2-3 % high on a cold box)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', default='ssd_300_vgg16_voc')
    ap.add_argument('--batch', type=int, default=32, help='per-GPU batch (weak scaling)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra-legs', action='store_true', help='skip roofline_hbm / per_config (they run at N = 1 only)')
    ap.add_argument('--eval-steps', type=int, default=5)
    ap.add_argument('--sync-bn', action='store_true',
                    help='N > 1: synchronise the pyramid tail\'s BatchNorm statistics over the ranks (detection/init.py:85 convert_syncbn_model); '
                         'default is local statistics, the documented local-BN mode of SURVEY.md 8e')
    ap.add_argument('--fast-mode-only', action='store_true', help='print only the fast_mode block (opt-in split-bf16 head GEMM), N = 1')
    ap.add_argument('--step-fn-only', action='store_true', help='print only the step_fn block (detection.init with and without graph_hot_path), N = 1')
    ap.add_argument('--rendezvous-only', action='store_true',
                    help='ranks form the process group, all-reduce their rank numbers, rank 0 prints a JSON line; no GPU work (launcher test)')
    args = ap.parse_args()

    # --gpus N > 1 without an outer torchrun: start the N ranks ourselves (nothing above this line has touched a GPU)
    _launch.self_launch_if_needed(args.gpus, __file__)

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus != world:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    backend = os.environ.get('SSDK_BENCH_BACKEND', 'nccl')   # nccl == RCCL on ROCm
    if args.rendezvous_only:
        import torch.distributed as dist
        if world > 1:
            dist.init_process_group(backend if backend != 'nccl' or torch.cuda.is_available() else 'gloo')
        t = torch.tensor([float(rank)], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({'rendezvous_only': True, 'n_gpus': world, 'ranks': dist.get_world_size() if world > 1 else 1,
                              'backend': dist.get_backend() if world > 1 else None, 'rank_sum': float(t.item())}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    if os.environ.get('SSDK_BENCH_ONE_GPU'):   # rehearsal of the N > 1 code path on a one-GPU box (with SSDK_BENCH_BACKEND=gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    rccl_ranks = 1
    if world > 1:
        import torch.distributed as dist
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)
        rccl_ranks = dist.get_world_size()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    _pygc.disable()   # (see quiesce(): garbage is collected between the timed regions, never inside one)
    if args.fast_mode_only:
        print(json.dumps({'fast_mode': fast_mode_legs(device, args.config, args.batch)}), flush=True)
        return
    if args.step_fn_only:
        print(json.dumps({'step_fn': step_fn_legs(device)}), flush=True)
        return
    hp = HotPath(args.config, args.batch, device)
    bn_modules = [m for m in (hp.extras, hp.tower, hp.neck) if m is not None]
    sync_bn = bool(args.sync_bn and world > 1 and bn_modules)
    if sync_bn:
        from single_shot_detection_amd.distributed import convert_sync_batchnorm
        for m in bn_modules:
            convert_sync_batchnorm(m)
    for _ in range(args.warmup):
        hp.train_step(world)
    # the cyclic collector stays out of the timed region (collected just before it instead): a generation-2 pass over the garbage of the
    # warm-up steps is several ms of host time -- more than the host runs ahead of the GPU in a 3.2 ms step
    import gc
    gc.collect()
    gc.disable()
    try:
        quiesce()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            hp.train_step(world, timed=True)
        barrier()
        dt = time.perf_counter() - t0
    finally:
        pass   # (the collector stays off for the rest of the run: quiesce() collects between the timed regions)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * args.batch * args.steps / dt

    # forward head GEMMs (igemm_streamk_kernel, ONE grouped launch per step) timed with events inside the timed region
    fwd_ms, fl_img = head_gemm_time(hp)
    flops_step = fl_img * args.batch
    achieved = flops_step / (fwd_ms * 1e-3) / 1e12

    # eval leg: heads forward + postprocess (NMS boxes/s = candidates entering NMS per second)
    hp.set_training(False)
    for _ in range(2):
        hp.eval_step()
    quiesce()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.eval_steps):
        hp.eval_step()
    barrier()
    dte = (time.perf_counter() - t0) / args.eval_steps
    cand = int(hp.post.last_nms_candidates.sum().item())
    # postprocess alone
    scores, locs = hp.forward_heads()
    scores, locs = scores.detach(), locs.detach()
    quiesce()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.eval_steps):
        hp.post.postprocess_padded((scores, locs), hp.anchors)
    barrier()
    dtp = (time.perf_counter() - t0) / args.eval_steps
    # the same on a "trained-like" score distribution (background logit + 6, about 1 % of the (anchor, class) pairs
    # pass the 0.01 threshold): the heads' random-init output above is the worst case (every pair passes)
    tl = scores.clone().view(args.batch, -1, hp.C)
    if hp.cfg['score_converter'] == 'SOFTMAX':
        tl[..., 0] += 6.0
    else:
        tl -= 4.6
    tl = tl.view(args.batch, -1)
    hp.post.postprocess_padded((tl, locs), hp.anchors)
    cand_tl = int(hp.post.last_nms_candidates.sum().item())
    quiesce()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.eval_steps):
        hp.post.postprocess_padded((tl, locs), hp.anchors)
    barrier()
    dtp_tl = (time.perf_counter() - t0) / args.eval_steps

    # a stream-K owner that gave up on a parked partial tile fills the tile with NaN and counts here: never reported as a result
    from single_shot_detection_amd import _lib as _ssdk_lib
    sk_timeouts = _ssdk_lib.streamk_timeouts()
    if world > 1:
        t = torch.tensor([float(sk_timeouts)], dtype=torch.float64, device=device)
        dist.all_reduce(t)
        sk_timeouts = int(t.item())
    if sk_timeouts:
        raise SystemExit(f'bench.py: {sk_timeouts} stream-K fix-up wait(s) timed out -- the head GEMM output is invalid, no line is printed')

    if rank == 0:
        A = hp.anchors.shape[0]
        ex_timing = hp.exchange.exchange_timing() if hp.exchange is not None else []
        out = {
            'metric': 'images/sec (train step) + NMS boxes/sec, SSD-300 VGG16 batch 32 @1/2/4/8 GPU',
            'value': value, 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'{args.config}: ' + ('MLFPN neck (8 TUMs + SFAM) + ' if hp.neck is not None else '') +
                                   ('extras conv-BN-ReLU + ' if hp.extras is not None else '') + ('shared-conv tower + ' if hp.tower is not None else '') +
                                   f'heads fwd+bwd (fp32 MFMA) + IoU-match + ' + ('HNM' if hp.cfg['loss'] == 'ce_hnm' else 'naive sampler') +
                                   f'/multibox loss fwd+bwd + SGD on the head-side params; backbone taps N(0,1) NHWC at the probed shapes, C={hp.C}, A={A}, G~U{{1..8}}',
                       'global_batch': world * args.batch, 'per_gpu_batch': args.batch, 'parallelism': f'dp{world}', 'sync_bn': sync_bn},
            'rccl_ranks': rccl_ranks, 'collective_backend': (backend if world > 1 else None), 'streamk_timeouts': sk_timeouts,
            'grad_bucket_bytes': {'heads': hp.bucket_heads.nbytes, 'rest': hp.bucket_rest.nbytes if hp.bucket_rest is not None else 0},
            'exchange': (None if hp.exchange is None else
                         {'what': 'distributed.BucketedDataParallel (the wrapper detection.init(distributed=True) uses): bucket indices in the order their '
                                  'all-reduces started in the last step, those started from a gradient hook (before the backward pass ended), and the '
                                  'head gradients that had to be copied into their bucket slots (0 = the kernels wrote them there)',
                          'start_order': hp.exchange.start_order, 'started_early': hp.exchange.started_early,
                          'heads_copied': int(getattr(hp.bucket_heads, 'copied_last', -1)),
                          # of the LAST timed step, per bucket (heads first): ring_ms = all-reduce started -> joined on the device timeline (an
                          # upper bound of the ring itself: the joining stream may have had work queued); exposed_ms = how long the join at
                          # the end of backward() stalled that stream (gloo: the host) -- 0 means the ring was hidden under the tail's backward
                          'buckets': ex_timing, 'ring_ms': [b['ring_ms'] for b in ex_timing], 'exposed_ms': [b['exposed_ms'] for b in ex_timing],
                          'recovered_steps': hp.exchange.recovered_steps}),
            'nms_boxes_per_sec': world * cand / dtp, 'postprocess_images_per_sec': world * args.batch / dtp,
            'eval_images_per_sec': world * args.batch / dte, 'nms_candidates_per_image': cand / args.batch,
            'postprocess_trained_like': {'images_per_sec': world * args.batch / dtp_tl, 'nms_boxes_per_sec': world * cand_tl / dtp_tl,
                                         'nms_candidates_per_image': cand_tl / args.batch, 'ms_per_batch': dtp_tl * 1e3},
            'roofline': {'bound': 'mfma', 'kernel': 'igemm_streamk_kernel (forward head GEMMs, all pyramid levels in one grouped stream-K launch of 512 persistent workgroups)',
                         'achieved': achieved, 'peak': PEAK_FP32_MATRIX_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved / PEAK_FP32_MATRIX_TFLOPS,
                         'traffic': measured_traffic(f'{args.config}:b{args.batch}:igemm_fwd_heads'),
                         'algorithmic_gflop_per_step': flops_step / 1e9, 'ms_per_step': fwd_ms},
        }
        if world == 1 and not args.no_extra_legs:
            del scores, locs, tl
            torch.cuda.empty_cache()
            hp.set_training(True)
            out['deterministic'] = deterministic_leg(hp, max(4, args.steps // 2))
            hp.set_training(False)
            out['roofline_hbm'] = hbm_legs(device)
            torch.cuda.empty_cache()
            # ... and at the largest single-GPU configuration (RetinaNet-500, batch 32: A = 47 961, 491 MB of logits, 63 MB of match
            # traffic): where the legs stop being launch-bound
            out['roofline_hbm_largest'] = hbm_legs(device, 'retina_rn50_500_coco', 32)
            torch.cuda.empty_cache()
            out['per_config'] = per_config_legs(device)
            out['serving'] = serving_legs(device)
            out['train_graph'] = train_graph_legs(device)
            out['step_fn'] = step_fn_legs(device)
            out['fast_mode'] = fast_mode_legs(device)
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(hp)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
