"""GPU parity of the conv-BN-ReLU compositions (SURVEY.md 8a H2 extras, H3 RetinaNet tower, 8f f1 FPN / TUM / SFAM) against fixtures that
the REFERENCE's own modules produced (tests/golden/blocks_small.npz, written by `tools/gen_golden.py --only blocks` from
bf/modules/conv.py:4-85, detection/detector_builder.py:57-109, detection/modules/predictors.py:8-76, bf/modules/features.py:52-300 on the
cases of tests/blocks_cases.py).  Both sides run the same harness (blocks_cases.run_case): eval() forward + backward, then one train()
step; compared are the outputs, the input gradients, every parameter gradient and every BatchNorm buffer after the step.

Tolerance: north_star asks for 1e-4 in fp32; these compositions are held to 2e-5 -- |diff| <= 2e-5 * (|ref| + max|ref|) for every
element of every output, gradient and buffer, no exceptions (measured on MI355X, tools/blocks_report.py: the worst element of any case is
at 3.8e-6 on that scale, train()-mode statistics over 4 rows included).  The absolute floor is relative to the tensor's largest magnitude
because fp32 GEMMs sum in another order than torch's CPU kernels: an element that is a near-cancellation of K products cannot be held to
a fraction of ITSELF."""
import os
import types

import numpy as np
import pytest
import torch

import blocks_cases
from conftest import GOLDEN
from single_shot_detection_amd.bf.modules import conv, features
from single_shot_detection_amd.detection import detector_builder
from single_shot_detection_amd.detection.modules import predictors

pytestmark = pytest.mark.gpu

MODS = types.SimpleNamespace(Conv2dBn=conv.Conv2dBn, DepthwiseConv2dBn=conv.DepthwiseConv2dBn, get_extras=detector_builder.get_extras,
                             SharedConvPredictor=predictors.SharedConvPredictor, FeaturePyramid=features.FeaturePyramid,
                             ThinnedUshapeModule=features.ThinnedUshapeModule,
                             ScalewiseFeatureAggregationModule=features.ScalewiseFeatureAggregationModule)

BAR = 2e-5


@pytest.fixture(scope='module')
def golden_blocks():
    return np.load(os.path.join(GOLDEN, 'blocks_small.npz'))


@pytest.mark.parametrize('case', sorted(blocks_cases.CASES))
def test_block_vs_reference_golden(case, golden_blocks):
    got = blocks_cases.run_case(case, MODS, torch.device('cuda'))
    want_keys = sorted(k for k in golden_blocks.files if k.startswith(case + '/'))
    assert sorted(got) == want_keys, (sorted(set(want_keys) ^ set(got))[:10])   # same outputs, same state_dict names, same sampling
    worst = {}
    for key in want_keys:
        ref, val = golden_blocks[key], got[key]
        bar = BAR
        if key.endswith('__shape'):
            assert np.array_equal(ref, val), key
        elif '/buffers/' in key and key.endswith('num_batches_tracked'):
            assert np.array_equal(ref, val), (key, ref, val)
        elif key.endswith('__sum_l2'):
            l2 = float(ref[1])
            assert abs(val[1] - ref[1]) <= bar * l2 + 1e-12, (key, val, ref)
            assert abs(val[0] - ref[0]) <= 10 * bar * l2 + 1e-12, (key, val, ref)
        else:
            scale = float(np.abs(ref).max()) if ref.size else 0.0
            err = np.abs(val.astype(np.float64) - ref.astype(np.float64))
            tol = bar * (np.abs(ref) + scale) + 1e-12
            bad = err > tol
            worst[key] = float((err / tol).max()) if err.size else 0.0
            assert not bad.any(), (key, int(bad.sum()), float(err.max()), scale, bar)
