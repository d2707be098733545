#!/usr/bin/env python3
"""Pretty-print a bench.py JSON line:  python tools/show_bench.py gpurun_out/x.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('value %.1f %s  ms/step %.3f  roofline %.1f %s frac %.3f' % (d['value'], d['unit'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['unit'], d['roofline']['frac']))
print('post worst: %.0f img/s  %.3g boxes/s | trained: %.0f img/s  %.3f ms | eval %.0f img/s' % (d['postprocess_images_per_sec'], d['nms_boxes_per_sec'], d['postprocess_trained_like']['images_per_sec'], d['postprocess_trained_like']['ms_per_batch'], d['eval_images_per_sec']))
for k, v in d.get('roofline_hbm', {}).get('legs', {}).items():
    print('  %-26s %8.1f us  %7.1f GB/s  frac %.3f' % (k, v['us'], v['achieved'], v['frac']))
for r in d.get('per_config', []):
    print('  %-24s b%-3d %7.3f ms/step %9.1f img/s  gemm %6.1f TF (%.3f)  post %9.0f img/s' % (r['config'], r['per_gpu_batch'], r['ms_per_step'], r['images_per_sec'], r['head_gemm_tflops'], r['head_gemm_frac'], r['postprocess_worst_case_images_per_sec']))
if 'cpu_baseline' in d:
    print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
for r in d.get('serving', []):
    print('  serving %-22s b%-3d eager %8.0f img/s   graph replay %8.0f img/s' % (r['config'], r['batch'], r['eager_images_per_sec'], r['graph_images_per_sec']))
for r in d.get('train_graph', []):
    print('  train step %-22s b%-3d eager %7.3f ms   graph replay %7.3f ms' % (r['config'], r['batch'], r['eager_ms_per_step'], r['graph_ms_per_step']))
