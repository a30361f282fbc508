"""profiles/rNN_traffic.json from the two PMC passes of tools/collect_profiles.cmd: HBM bytes per launch of the two
dominant kernels at bench.py's workload (max over dispatches = the full-size launches; FETCH_SIZE x2 per the gfx950
correction for 16-B/lane streaming reads, MI355X_MICROARCH.md HBM section).
usage: python tools/make_traffic_json.py <fetch counter_collection.csv> <write counter_collection.csv> <round>"""
import csv, json, sys


def peak(path, counter, kernel):
    best = 0.0
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter and kernel in row["Kernel_Name"]:
                best = max(best, float(row["Counter_Value"]))
    return best


fetch, write, rnd = sys.argv[1:4]
N, D = 1_000_000, 100
cmd = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 5 --warmup 2 "
       "--no-cpu-baseline --no-hvp --no-block --no-fp32 --no-c3")
out = {"workload": {"rows": N, "dim": D}, "command": cmd}
for kern, alg, note in (("k_trsm_rows_small", 16.0 * N * D + 8.0 * N, "16ND + 8N"),
                        ("k_glm_hess_tile", 8.0 * N * D + 24.0 * N + 8.0 * D * D, "8ND + 24N + 8D^2")):
    f_kb, w_kb = peak(fetch, "FETCH_SIZE", kern), peak(write, "WRITE_SIZE", kern)
    out[kern] = {"fetch_size_kb_raw": f_kb, "fetch_correction": 2.0, "write_size_kb": w_kb,
                 "hbm_bytes_per_launch": (2.0 * f_kb + w_kb) * 1000.0,
                 "algorithmic_bytes": alg, "algorithmic_formula": note,
                 "source": "profiles/%s_pmc_hbm_traffic.txt (max over dispatches = the full-size launches)" % rnd}
print(json.dumps(out, indent=1))
