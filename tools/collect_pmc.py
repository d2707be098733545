#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc passes -> profiles/rNN_pmc.json.

  python tools/collect_pmc.py <out.json> <config:bN> <dir-with-counter_collection-csvs>...

Every directory is one rocprofv3 run (`--kernel-trace --pmc <counters>`); counters are averaged per launch over the launches of a
kernel (the first launch of each kernel, which includes cold caches, is dropped when there are more than two)."""
import collections, csv, glob, json, sys

KEYS = {   # json key suffix -> substring of the kernel name
    'igemm_fwd_heads': 'igemm_streamk_kernel',
    'igemm_fwd_heads_tiles': 'igemm_dma_kernel<false, false, false, 4',
    'igemm_scatter_dgrad': 'igemm_dma_kernel<false, false, true, 4',
    'igemm_wgrad': 'igemm_wgrad_dma_kernel', 'wgrad_rows': 'igemm_wgrad_rows_kernel', 'anchor_rowgemm': 'anchor_rowgemm_kernel',
    'anchor_dx': 'anchor_dx_kernel', 'gather_rows': 'gather_rows_kernel', 'anchor_plan': 'anchor_plan_kernel', 'reduce_partials': 'reduce_partials_kernel',
    'loss_bwd': 'loss_bwd_kernel', 'loss_fwd': 'loss_fwd_kernel', 'hnm_rows': 'hnm_rows_kernel', 'hnm_select': 'hnm_select_kernel',
    'pack_dy': 'pack_dy_kernel', 'assign': 'assign_kernel', 'gt_argmax': 'gt_argmax_kernel',
    'post_select': 'post_select2_kernel', 'post_nms': 'post_nms_wave_kernel', 'post_merge': 'post_merge2_kernel', 'post_tau': 'post_tau_kernel', 'post_finish': 'post_finish_kernel',
}
out_path, tag, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in dirs:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            for key, sub in KEYS.items():
                if sub in r['Kernel_Name']:
                    acc[key][r['Counter_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            for key, sub in KEYS.items():
                if sub in r['Kernel_Name']:
                    dur[key].append((int(r['Dispatch_Id']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
try:
    res = json.load(open(out_path))
except Exception:
    res = {}
for key, counters in acc.items():
    e = {'kernel': KEYS[key]}
    for name, vals in counters.items():
        vals = [v for _, v in sorted(vals)]
        if len(vals) > 2:
            vals = vals[1:]
        mean = sum(vals) / len(vals)
        if name in ('FETCH_SIZE', 'WRITE_SIZE'):
            e[name + '_KiB'] = mean
        elif name == 'GRBM_GUI_ACTIVE':
            e['cycles_per_launch'] = mean / 8.0   # the counter is summed over the 8 XCDs
        else:
            e[name] = mean
    if dur[key]:
        ds = [v for _, v in sorted(dur[key])]
        e['profiled_us'] = sum(ds[1:] if len(ds) > 2 else ds) / max(1, len(ds[1:] if len(ds) > 2 else ds)) / 1e3
    res[f'{tag}:{key}'] = e
res['_note'] = ('rocprofv3 --kernel-trace --pmc <X> -- python3 bench.py ... (separate passes for FETCH_SIZE, WRITE_SIZE and the SQ/GRBM counters); per-launch '
                'means. On gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads: double it (MI355X_MICROARCH.md, HBM section). '
                'cycles_per_launch = GRBM_GUI_ACTIVE / 8; SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs.')
json.dump(res, open(out_path, 'w'), indent=1, sort_keys=True)
print('wrote', out_path, 'keys', sorted(k for k in res if k.startswith(tag)))
