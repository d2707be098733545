"""Wiring + step_fn -- mirror of detection/init.py:19-137 (the only seam between the training runtime and the
detection hot path).  Same signature and return triple; everything behind step_fn runs on the GPU through libssdk:
anchors are device-resident, the target never leaves the device (init.py:115's H2D copy is gone) and the
postprocessor no longer falls back to the host."""
import functools
import logging

import torch

from ..bf import base as base_builder
from ..utils import filter_kwargs
from . import detector_builder
from . import sampler as _sampler_mod
from .box_coder import BoxCoder
from .detector_wrapper import DetectorWrapper
from .losses.multibox_loss import MultiboxLoss
from .postprocessor import Postprocessor
from .target_assigner import TargetAssigner


def init(device, model_args, box_coder_args, postprocess_args, loss_args, sampler_args, target_assigner_args, state={},
         preprocess=None, parallel=False, distributed=False, graph_hot_path=False, graph_gt_capacity=None):
    """detection/init.py:19-137.  ``graph_hot_path`` (not in the reference; default off): the libssdk part of a TRAINING step -- pyramid
    tail + heads forward, target assignment, sampler + multibox loss and their backward -- is captured once per input shape as two HIP
    graphs (graphs.GraphedSegment) and replayed; the backbone stays eager PyTorch.  The step then costs the host two graph launches
    instead of ~90 kernel launches (the eager step needs ~2.3 ms of host time per 3 ms GPU step at SSD-300 / batch 32, and is host
    bound at small batches).  Same numbers as the eager step (bit for bit where the kernels are deterministic).  Limits: fixed batch
    size and image size (a new shape is captured again), at most ``graph_gt_capacity`` ground-truth boxes per batch (default: 64 per
    image), not with ``distributed=True`` (parameter gradients bypass autograd's hooks), the predictor's parameter gradients live in
    static buffers that the next step overwrites."""
    assert not (parallel and distributed)
    assert not (graph_hot_path and distributed), 'graph_hot_path: the exchange wrapper starts its rings from gradient hooks, which a captured backward does not fire'
    if parallel:
        raise NotImplementedError('single-process DataParallel is not reproduced; use one process per GPU (distributed=True)')
    if 'model' in state:
        detector = state.pop('model')
    else:
        base = base_builder.create_base(**model_args['base'])
        detector = filter_kwargs(detector_builder.build)(base, anchor_generator_params=model_args['anchor_generator'],
                                                         **model_args['detector'])
        if 'model_dict' in state:
            detector.load_state_dict(state.pop('model_dict'))
    detector = detector.to(device).to(memory_format=torch.channels_last)
    if distributed:
        # one process per GPU; gradients of the predictor are averaged over RCCL/xGMI (init.py:80-86 used apex DDP + SyncBN): the flat
        # two-bucket exchange bench.py --gpus N measures (distributed.BucketedDataParallel: the heads' ring starts as soon as the heads'
        # backward has run, under the backward pass of the pyramid tail and the backbone; zero-copy where the weight-gradient kernels
        # write into the bucket).  Hot-path BatchNorms stay on libssdk with their statistics all-reduced (convert_sync_batchnorm).
        from ..distributed import BucketedDataParallel, convert_sync_batchnorm
        detector = convert_sync_batchnorm(detector)
        detector.predictor = BucketedDataParallel(detector.predictor)
    # (the pyramid tail's weight gradients in one grouped launch per backward pass -- ops.defer_weight_gradients -- is an opt-in of the
    # caller: it bypasses autograd's AccumulateGrad, which hooks, autograd.grad and DistributedDataParallel rely on)
    logging.info(detector)

    sampler = getattr(_sampler_mod, sampler_args['name'])
    kwargs = {k: v for k, v in sampler_args.items() if k in sampler.__code__.co_varnames and not k.startswith('_')}
    sampler = functools.partial(sampler, **kwargs)

    box_coder = BoxCoder(**box_coder_args)
    criterion = MultiboxLoss(sampler=sampler, box_coder=box_coder, **loss_args)
    postprocessor = Postprocessor(box_coder, **postprocess_args)
    target_assigner = TargetAssigner(**target_assigner_args)
    detector_wrapper = DetectorWrapper(detector, preprocess, postprocessor)

    def init_epoch_state():
        return {'class_loss': 0.0, 'loc_loss': 0.0, 'loss': 0.0}

    class _HotSegment(object):
        """graph_hot_path: everything behind the backbone for ONE input geometry -- the captured segment, its static ground-truth buffers
        and the (cached, device-resident) anchors."""

        def __init__(self, imgs, taps, n_sources, x_index, ground_truth):
            from ..graphs import GraphedSegment
            from .target_assigner import PackedGroundTruth
            B = imgs.shape[0]
            cap = int(graph_gt_capacity) if graph_gt_capacity else 64 * B
            self.packed = PackedGroundTruth(torch.zeros((cap, 6), dtype=torch.float32, device=device),
                                            torch.zeros((B + 1,), dtype=torch.int32, device=device))
            self.packed.update_(ground_truth)   # (the first batch's boxes are in the buffers while the segment warms up and is captured)
            img_like = torch.empty((0,) + tuple(imgs.shape[1:]), device=device)   # (anchors only need the image's size and device)
            self.priors = None

            def fn(*t):
                scores, locs, loc_sources = detector.predictor.forward_from_taps(list(t[:n_sources]), t[x_index])
                priors = detector.generate_anchors(img_like, loc_sources)
                self.priors = priors
                target = target_assigner.encode_ground_truth(self.packed, priors)
                loss, class_loss, loc_loss = criterion((scores, locs), priors, target)
                return loss, class_loss, loc_loss, scores, locs
            params = [p for n, p in detector.predictor.named_parameters() if not n.startswith('features.')]
            self.segment = GraphedSegment(fn, taps, params)

    segments = {}

    def step_fn(step, phase, batch, state):
        imgs, ground_truth = batch
        imgs = imgs.to(device, non_blocking=True)
        if graph_hot_path and phase == 'train' and detector.training and torch.is_grad_enabled():
            sources, x = detector.predictor.features(imgs)     # the backbone (and a neck inside ``features``): eager PyTorch
            sources = list(sources)
            x_index = next((i for i, s_ in enumerate(sources) if s_ is x), len(sources))
            taps = [t.contiguous(memory_format=torch.channels_last) for t in (sources if x_index < len(sources) else sources + [x])]
            key = (tuple(imgs.shape), tuple(tuple(t.shape) for t in taps))
            hot = segments.get(key)
            if hot is None:
                hot = segments[key] = _HotSegment(imgs, taps, len(sources), x_index, ground_truth)
            else:
                hot.packed.update_(ground_truth)
            loss, class_loss, loc_loss, scores, locs = hot.segment(*taps)
            prediction, priors = [scores, locs], hot.priors
        else:
            *prediction, priors = detector(imgs)
            target = target_assigner.encode_ground_truth(ground_truth, priors)
            loss, class_loss, loc_loss = criterion(prediction, priors, target)
        prediction = [x.detach() for x in prediction]
        if phase == 'eval':
            prediction = postprocessor.postprocess(prediction, priors)
        if step == 0:
            [setattr(step_fn, attr, 0.0) for attr in ['class_loss', 'loc_loss', 'loss']]
        step_fn.class_loss += class_loss.item()   # init.py:127-128: the two D2H syncs of the reference are kept
        step_fn.loc_loss += loc_loss.item()
        step_fn.loss = step_fn.class_loss + step_fn.loc_loss
        state['class_loss'] = step_fn.class_loss / (step + 1)
        state['loc_loss'] = step_fn.loc_loss / (step + 1)
        state['loss'] = step_fn.loss / (step + 1)
        return loss, prediction, state

    step_fn.hot_segments = segments   # (graph_hot_path: the captured segments by input geometry -- diagnostics, bench.py)
    return detector_wrapper, init_epoch_state, step_fn
