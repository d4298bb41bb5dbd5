"""Aggregate a rocprofv3 --pmc pass (SQ counters) per kernel: sums over all dispatches.
usage: pmc_sq.py counter_collection.csv out.csv"""
import csv, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); names = set()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0]
    tot[k][r['Counter_Name']] += float(r['Counter_Value']); names.add(r['Counter_Name'])
    key = (r['Dispatch_Id'])
    if key not in seen:
        seen.add(key); calls[k] += 1
names = sorted(names)
w = csv.writer(open(sys.argv[2], 'w'))
w.writerow(['kernel', 'dispatches'] + names)
for k in tot:
    w.writerow([k, calls[k]] + [int(tot[k][n]) for n in names])
for k in tot:
    if 'gemm' in k or 'mfma_peak' in k:
        m = tot[k].get('SQ_INSTS_VALU_MFMA_MOPS_F64', 0.)
        print('%-40s dispatches %5d  MFMA wave-instr (MOPS_F64) %.4g -> %.4g flop executed' % (k[:40], calls[k], m, m * 512))
