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
         preprocess=None, parallel=False, distributed=False):
    assert not (parallel and distributed)
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

    def step_fn(step, phase, batch, state):
        imgs, ground_truth = batch
        imgs = imgs.to(device, non_blocking=True)
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

    return detector_wrapper, init_epoch_state, step_fn
