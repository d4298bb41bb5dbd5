"""Aggregate two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of
`bench.py --streams 1 --steps 1 --warmup 0 --no-cpu-baseline`) into profiles/<name>.json.
FETCH_SIZE / WRITE_SIZE are reported in KB; FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section:
on gfx950 it counts 16-B-per-lane loads at half their bytes).
Every update launch of the run is counted (setup, warm-up, timed steps, stage pass: all the same kernels on the
same shapes), traffic is reported PER LAUNCH like the roofline's flops.
usage: pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv objects_per_launch out.json [kernel substring]
(kernel substring: default k_gemm_nt; k_chol_dag for the single-launch factorisation of `bench.py --batch 1 --streams 1`)"""
import csv, json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
fetch_csv, write_csv, objs, out = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
KSUB = sys.argv[5] if len(sys.argv) > 5 else 'k_gemm_nt'

def load(path, counter):
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter: continue
        rows[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    tot = collections.defaultdict(float); calls = collections.defaultdict(int)
    for name, vals in rows.items():
        if KSUB in name and KSUB != 'k_gemm_nt': print(counter, name, [round(v / 1e6, 2) for v in vals], file=sys.stderr)
        # a counter is reported once per XCD partition of a dispatch on some rocprofv3 versions: dispatches are what we count,
        # and with several objects per launch the run also holds the ONE-object launch of bench.py's set-up (same kernel name,
        # a fraction of the traffic): launches below half of the largest are left out of the per-launch average
        if objs > 1 and KSUB in name and KSUB != 'k_gemm_nt':
            top = max(vals)
            vals = [v for v in vals if v >= 0.5 * top]
        tot[name] = sum(vals); calls[name] = len(vals)
    return tot, calls

f, fc = load(fetch_csv, 'FETCH_SIZE')
w, wc = load(write_csv, 'WRITE_SIZE')
gemm = [k for k in f if KSUB in k]
launches = sum(fc[k] for k in gemm)
wl = sum(wc[k] for k in w if KSUB in k)
fetch_b = 2. * 1024. * sum(f[k] for k in gemm) / launches
write_b = 1024. * sum(w[k] for k in w if KSUB in k) / wl
asm = [k for k in f if 'k_assemble_train' in k]
na = max(1, sum(fc[k] for k in asm))
res = dict(source='rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --no-extras`; all update '
                  'launches of the run averaged; FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section (16-B-per-lane '
                  'loads on gfx950)',
           kernel_sources_sha256=bench.kernel_source_hash(),
           kernel=KSUB + ' (every launch of the run)', objects_per_launch=objs,
           launches_in_run=launches, fetch_bytes_per_launch=fetch_b, write_bytes_per_launch=write_b,
           traffic_bytes_per_launch=fetch_b + write_b,
           fetch_bytes_per_fit=fetch_b * (63 if KSUB == 'k_gemm_nt' else 1) / objs, write_bytes_per_fit=write_b * (63 if KSUB == 'k_gemm_nt' else 1) / objs,
           assemble_train_write_bytes_per_launch=1024. * sum(w[k] for k in w if 'k_assemble_train' in k) / na,
           assemble_train_fetch_bytes_per_launch=2. * 1024. * sum(f[k] for k in asm) / na,
           per_kernel_KB_total_in_run=dict(FETCH_SIZE={k: f[k] for k in f}, WRITE_SIZE={k: w[k] for k in w}),
           calls_in_run={k: fc[k] for k in fc})
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps({k: res[k] for k in ('launches_in_run', 'fetch_bytes_per_launch', 'write_bytes_per_launch', 'traffic_bytes_per_launch', 'fetch_bytes_per_fit')}))
