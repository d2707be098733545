import json, sys, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda:0')
for cfg, b in (('ssd_300_vgg16_voc', 64), ('retina_rn50_500_coco', 32)):
    r = bench.hbm_legs(dev, cfg, b)
    print(r['workload'], 'boundary %.2f us' % r['kernel_boundary_us'])
    for k, v in r['legs'].items():
        print('  %-26s %8.1f us  frac %.3f  launches %d  floor %6.1f us (%s)  of_floor %.2f' % (k, v['us'], v['frac'], v['launches'], v['floor_us'], v['bound'], v['of_floor']))
    torch.cuda.empty_cache()
