"""The one rank of tests/test_end_to_end_gpu.py::test_rccl_world_of_one: everything of the N > 1 exchange step that can run on a one-GPU box
with the REAL backend ('nccl' = RCCL on ROCm; the role of bf/training/env.py:55-67 + the apex DDP all-reduce of detection/init.py:80-86):
init_process_group('nccl', device_id=...), the ReduceOp.AVG probe, the head-gradient bucket (36 MB at SSD-300 / 81 classes) forced through
dist.all_reduce with the zero-copy views attached, a second (async, overlapped) round like bench.py's two-phase step, destroy.  Started as a
fresh child process by single_shot_detection_amd.launch (which sets HSA_ENABLE_IPC_MODE_LEGACY=0 and the rendezvous variables)."""
import json
import os
import sys

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import bench  # noqa: E402
from single_shot_detection_amd import _lib, distributed  # noqa: E402


def main():
    out_path = sys.argv[1]
    assert os.environ['WORLD_SIZE'] == '1' and os.environ['RANK'] == '0'
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', device_id=dev)
    res = {'backend': dist.get_backend(), 'world': dist.get_world_size(), 'ipc_legacy_env': os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}
    res['avg_supported'] = bool(distributed._avg_supported(None, dev))
    hp = bench.HotPath('ssd_300_vgg16_voc', 2, dev)
    hp.enable_exchange()                 # the N > 1 wrapper (inactive at world size 1: its buckets are driven by hand below)
    hp.train_step()
    hp.train_step()
    torch.cuda.synchronize()
    bucket = hp.bucket_heads
    res['bucket_bytes'] = bucket.nbytes
    # after a step every head gradient IS its bucket slot (zero-copy): the collective below moves nothing first
    res['views_alias_bucket'] = all(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket.views))
    before = bucket.flat.clone()
    bucket.start_(force=True)            # ReduceOp.AVG over a group of one: values unchanged, but the 36 MB really go through RCCL
    side = torch.randn((1024, 1024), device=dev) @ torch.randn((1024, 1024), device=dev)   # (work enqueued while the collective runs)
    bucket.finish_()
    torch.cuda.synchronize()
    res['copied_last'] = int(bucket.copied_last)
    res['max_abs_change'] = float((bucket.flat - before).abs().max())
    res['views_alias_after'] = all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket.views))
    res['finite'] = bool(torch.isfinite(bucket.flat).all()) and bool(torch.isfinite(side).all())
    if hp.bucket_rest is not None:
        hp.bucket_rest.allreduce_(force=True)
        res['rest_bytes'] = hp.bucket_rest.nbytes
    # a packed SyncBatchNorm buffer through the same backend (fp64 sum)
    sums = torch.arange(2 * 256 + 2, dtype=torch.float64, device=dev)
    dist.all_reduce(sums)
    res['sums_ok'] = bool(torch.equal(sums.cpu(), torch.arange(2 * 256 + 2, dtype=torch.float64)))
    hp.opt.step()
    torch.cuda.synchronize()
    res['streamk_timeouts'] = _lib.streamk_timeouts()
    dist.destroy_process_group()
    with open(out_path, 'w') as f:
        json.dump(res, f)


if __name__ == '__main__':
    main()
