"""HBM traffic of ONE PASS of a launch-per-step workload (config 4: 128 light curves per pass, ~20 launches) from two
rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench.py command).
Every kernel of the library counts; the number of passes in the run = the number of k_finish launches (one per pass).
FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: gfx950 tallies 16-B-per-lane
loads at half their bytes).
usage: pmc_pass_traffic.py fetch.csv write.csv objects_per_pass out.json"""
import csv, json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

fetch_csv, write_csv, objs, out = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]


def load(path, counter):
    tot, calls = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        k = r['Kernel_Name'].split('(')[0]
        tot[k] += float(r['Counter_Value'])
        calls[k] += 1
    return tot, calls


f, fc = load(fetch_csv, 'FETCH_SIZE')
w, wc = load(write_csv, 'WRITE_SIZE')
mine = lambda k: k.startswith('k_') or ' k_' in k            # the library's kernels (runtime fills / copies left out)
skip = ('k_mfma_selftest',)
passes = sum(c for k, c in fc.items() if 'k_finish' in k)
fetch_b = 2. * 1024. * sum(v for k, v in f.items() if mine(k) and k not in skip) / passes
write_b = 1024. * sum(v for k, v in w.items() if mine(k) and k not in skip) / passes
res = dict(source='rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of one bench.py command; every kernel of the '
                  'library summed, divided by the number of passes (= k_finish launches); FETCH_SIZE doubled per '
                  'MI355X_MICROARCH.md HBM section (16-B-per-lane loads on gfx950)',
           kernel_sources_sha256=bench.kernel_source_hash(), objects_per_pass=objs, passes_in_run=passes,
           fetch_bytes_per_pass=fetch_b, write_bytes_per_pass=write_b, traffic_bytes_per_pass=fetch_b + write_b,
           traffic_bytes_per_launch=fetch_b + write_b,
           fetch_bytes_per_fit=fetch_b / objs, write_bytes_per_fit=write_b / objs, traffic_bytes_per_fit=(fetch_b + write_b) / objs,
           per_kernel_bytes_per_pass={k: dict(fetch=2. * 1024. * f.get(k, 0.) / passes, write=1024. * w.get(k, 0.) / passes,
                                              launches_per_pass=fc.get(k, 0) / passes)
                                      for k in sorted(set(f) | set(w)) if mine(k) and k not in skip})
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps({k: res[k] for k in ('passes_in_run', 'fetch_bytes_per_fit', 'write_bytes_per_fit', 'traffic_bytes_per_fit')}))
