#!/bin/bash
# Per-call durations of the kernels of a config's training steps (the 22 largest by total time), from a rocprofv3 kernel trace:
# which calls carry the time?   bash tools/step_calls.sh [config] [batch]  (on the GPU box; output gpurun_out/step_calls_<config>.txt)
R=${GRAFT_REPO_ROOT:-/root/repo}
CFG=${1:-m2det_512_vgg16_coco}; B=${2:-16}
cd /tmp && export TMPDIR=/tmp
W=/tmp/step_calls; rm -rf $W; mkdir -p $W $R/gpurun_out
timeout -k 10 400 rocprofv3 --kernel-trace -d $W -o p -- python3 $R/bench.py --config $CFG --batch $B --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs > $W/run.log 2>&1
python3 - $W/p_results.db > $R/gpurun_out/step_calls_$CFG.txt <<'PY'
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = db.execute(f"select s.display_name, d.end - d.start, d.grid_size_x, d.grid_size_y from {disp} d join {sym} s on d.kernel_id = s.id").fetchall()
by = collections.defaultdict(list)
for name, du, gx, gy in rows:
    short = name.split('(')[0].replace('void ', '').replace('ssdk::', '')
    if 'at::native' in name: short = 'torch ' + ('add' if 'Functor_add' in name or 'OnSelf_add' in name else 'cat' if 'CatArray' in name else 'copy' if 'nocast' in name or 'copy' in name.lower() else 'sgd' if 'multi_tensor' in name else name[:60])
    by[short].append((du / 1e3, gx, gy))
tot_all = sum(c[0] for v in by.values() for c in v)
print(f'all kernels: {tot_all / 1e3:.1f} ms in {sum(len(v) for v in by.values())} launches')
for k in sorted(by, key=lambda k: -sum(c[0] for c in by[k]))[:22]:
    calls = by.get(k, [])
    if not calls: continue
    calls.sort(reverse=True)
    tot = sum(c[0] for c in calls)
    print(f'{k}: {len(calls)} calls, {tot / 1e3:.2f} ms; buckets by duration:')
    edges = [5, 10, 20, 50, 100, 200, 1e9]
    lo = 0
    for e in edges:
        sel = [c for c in calls if lo <= c[0] < e]
        if sel: print(f'   {lo:>5.0f}..{e if e < 1e9 else float("inf"):<6} us: {len(sel):4d} calls, {sum(c[0] for c in sel) / 1e3:7.2f} ms')
        lo = e
    print('   longest:', ', '.join(f'{c[0]:.0f}us(grid {c[1]}x{c[2]})' for c in calls[:8]))
PY
cat $R/gpurun_out/step_calls_$CFG.txt
