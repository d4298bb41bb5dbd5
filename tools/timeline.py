"""Developer analysis of a rocprofv3 --kernel-trace CSV: how much of the wall time has a GEMM running,
how much only leaf kernels (for the N-fits-in-flight mode)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    name = r['Kernel_Name']
    if 'k_mfma_peak' in name or 'k_stream_store' in name: continue
    kind = 'gemm' if 'k_gemm_nt' in name else ('potrf' if 'k_potrf' in name else ('trsm' if 'k_trsm' in name else 'other'))
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), kind))
ev.sort()
t0 = ev[len(ev) // 5][0]; t1 = ev[-len(ev) // 10][1]
pts = []
for s, e, k in ev:
    s = max(s, t0); e = min(e, t1)
    if e > s: pts.append((s, 1, k)); pts.append((e, -1, k))
pts.sort()
cnt = {'gemm': 0, 'potrf': 0, 'trsm': 0, 'other': 0}
last = t0; acc = {}
for t, d, k in pts:
    key = ('gemm x%d' % min(cnt['gemm'], 4)) if cnt['gemm'] else ('leaf only' if cnt['potrf'] + cnt['trsm'] else ('other' if cnt['other'] else 'idle'))
    acc[key] = acc.get(key, 0) + (t - last); last = t
    cnt[k] += d
tot = t1 - t0
print('window %.1f ms, %d kernels' % (tot / 1e6, len(ev)))
for k in sorted(acc): print('  %-10s %6.1f %%' % (k, 100. * acc[k] / tot))
