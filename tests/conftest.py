import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, 'tests', 'golden')
CONFIG_NAMES = ['ssd_mb2_voc', 'ssd_300_vgg16_voc', 'ssd_512_vgg16_coco', 'retina_rn50_500_coco', 'm2det_512_vgg16_coco']
GOLDEN_BATCH = {'ssd_mb2_voc': 2, 'ssd_300_vgg16_voc': 4, 'ssd_512_vgg16_coco': 2, 'retina_rn50_500_coco': 2,
                'm2det_512_vgg16_coco': 2}


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def kats():
    return np.load(os.path.join(GOLDEN, 'kats.npz'))


_cache = {}


def load_golden(name):
    if name not in _cache:
        _cache[name] = dict(np.load(os.path.join(GOLDEN, f'{name}.npz')))
    return _cache[name]


@pytest.fixture(scope='session')
def golden():
    return load_golden


def dense_from_rows(rows, vals, shape):
    out = np.zeros(shape, np.float32)
    if len(rows):
        out[rows[:, 0], rows[:, 1]] = vals
    return out


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(autouse=True)
def _streamk_waits_never_ran_out(request):
    """After every GPU test: no stream-K owner of ssdk_heads_fwd gave up on a parked partial tile (csrc/conv.hip; a timeout fills the
    tile with NaN, counts here and makes the next ssdk_heads_fwd fail -- it must never pass silently)."""
    yield
    if request.node.get_closest_marker('gpu') is None or not has_gpu():
        return
    from single_shot_detection_amd import _lib
    assert _lib.streamk_timeouts() == 0, 'a stream-K fix-up wait timed out during this test'


@pytest.fixture(autouse=True)
def _process_wide_switches_do_not_leak():
    """The two process-wide switches (ops.defer_weight_gradients, the heads' fast mode) are only ever set through scoped forms
    (ops.deferred_weight_gradients / heads.fast_mode context managers; bench.HotPath scopes deferral to its train_step): every test starts
    from the library's defaults, and a test that LEAVES one switched on fails here instead of changing what the next test measures."""
    from single_shot_detection_amd import _lib, ops
    ops.defer_weight_gradients(False)
    _lib.fast_mode = None
    built = __import__('os').path.exists(_lib.LIB_PATH)
    if built:
        ops.set_deterministic(False)
    yield
    leaked = (ops.defer_weight_gradients(False), _lib.fast_mode, ops.set_deterministic(False) if built else False)
    _lib.fast_mode = None
    assert leaked == (False, None, False), f'a process-wide switch leaked out of this test: (defer_weight_gradients, fast_mode, deterministic) = {leaked}'
