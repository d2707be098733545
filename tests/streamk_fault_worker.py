"""Child process of tests/test_heads_gpu.py::test_streamk_timeout_is_loud_in_graph_replays: a stream-K workgroup that never raises its
flag (fault injected through ssdk_debug_streamk_fault) must be LOUD everywhere -- the owner's tile is NaN in the replay that lost it, the
sticky host word is set, GraphedCallable refuses the next call, a raw replay stores NaN in EVERY tile (the kernel reads the workspace's
counter at entry: a replay never passes through ssdk_heads_fwd's host-side check, and would otherwise add the lost launch's stale partial
tile), and the eager entry point fails.  The poison is per process and sticky: hence the child process."""
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from single_shot_detection_amd import _lib, synthetic as syn  # noqa: E402
from single_shot_detection_amd.detection import detector_builder  # noqa: E402
from single_shot_detection_amd.detection.modules.heads import multi_level_heads  # noqa: E402
from single_shot_detection_amd.graphs import GraphedCallable  # noqa: E402


def main():
    out_path = sys.argv[1]
    cfg = syn.CONFIGS['ssd_300_vgg16_voc']
    levels, C, B = cfg['levels'], cfg['num_classes'], 32
    torch.manual_seed(5)
    heads = detector_builder.get_heads([l[0] for l in levels], [l[2] for l in levels], C).cuda()
    xs = [torch.randn((B, cin, h, h), device='cuda').contiguous(memory_format=torch.channels_last) for cin, h, _ in levels]

    def step():
        with torch.no_grad():
            return multi_level_heads(xs, xs, heads)[0]
    res = {}
    g = GraphedCallable(step, [], warmup=2)
    out = g()
    torch.cuda.synchronize()
    res['healthy_finite'] = bool(torch.isfinite(out).all())
    res['healthy_poisoned'] = _lib.streamk_poisoned()
    healthy = out.clone()
    lib = _lib.lib()
    nan_after_fault = 0
    for wg in (100, 101, 102, 103):   # (a workgroup whose range begins on a tile boundary parks nothing: try its neighbours)
        _lib.check(lib.ssdk_debug_streamk_fault(wg, 1 << 12), 'ssdk_debug_streamk_fault')
        g.graph.replay()
        torch.cuda.synchronize()
        nan_after_fault = int(torch.isnan(g.static_out).sum())
        if nan_after_fault:
            break
    res['dropped_workgroup'] = wg
    res['nan_in_faulted_replay'] = nan_after_fault
    res['untouched_rows_equal'] = bool((torch.isnan(g.static_out) | (g.static_out == healthy)).all())
    res['poisoned_after_fault'] = _lib.streamk_poisoned()
    res['timeouts'] = _lib.streamk_timeouts()
    _lib.check(lib.ssdk_debug_streamk_fault(-1, 0), 'ssdk_debug_streamk_fault')   # the fault is gone; the poison stays
    try:
        g()
        res['graphed_callable_raises'] = False
    except _lib.SsdkError:
        res['graphed_callable_raises'] = True
    g.graph.replay()
    torch.cuda.synchronize()
    res['all_nan_in_later_replay'] = bool(torch.isnan(g.static_out).all())
    try:
        step()
        res['eager_raises'] = False
    except (ValueError, _lib.SsdkError):
        res['eager_raises'] = True
    # recovery: ssdk_streamk_reset clears the workspace and the sticky word -- eager calls and the SAME captured graph are healthy again
    _lib.streamk_reset()
    res['poisoned_after_reset'] = _lib.streamk_poisoned()
    res['timeouts_after_reset'] = _lib.streamk_timeouts()
    out2 = g()
    torch.cuda.synchronize()
    res['replay_after_reset_equals_healthy'] = bool(torch.equal(out2, healthy))
    res['eager_after_reset_equals_healthy'] = bool(torch.equal(step(), healthy))
    json.dump(res, open(out_path, 'w'))


if __name__ == '__main__':
    main()
