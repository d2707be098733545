"""effective shader clock of the head GEMM in rocprofv3 --pmc GRBM_GUI_ACTIVE runs: python tools/clock_check.py <dir>..."""
import csv, glob, sys
for d in sys.argv[1:]:
    kt = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
    cc = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    dur = {r['Dispatch_Id']: int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(kt)) if ('igemm_dma_kernel<false, false, false' in r['Kernel_Name'] or 'igemm_fwd_kernel<4, false, false, false, false' in r['Kernel_Name'])}
    cyc = {r['Dispatch_Id']: float(r['Counter_Value']) for r in csv.DictReader(open(cc)) if r['Counter_Name'] == 'GRBM_GUI_ACTIVE' and r['Dispatch_Id'] in dur}
    ids = sorted(dur, key=int)[1:]
    print(d, 'launches', len(ids), 'mean ms %.3f' % (sum(dur[i] for i in ids) / len(ids) / 1e6), 'mean GHz %.3f' % (sum(cyc[i] / 8 / dur[i] for i in ids) / len(ids)),
          'cycles/launch %.3fM' % (sum(cyc[i] for i in ids) / 8 / len(ids) / 1e6))
