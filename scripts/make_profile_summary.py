"""Summarise the rocprofv3 passes of bench.py into profiles/<round>/search_kernel_summary.json.

Inputs (CSV output of rocprofv3, see scripts/profile_round.sh):
  <dir>/trace/...kernel_trace.csv        --kernel-trace --stats
  <dir>/pmc_fetch/...counter_collection.csv   --pmc FETCH_SIZE
  <dir>/pmc_write/...counter_collection.csv   --pmc WRITE_SIZE
  <dir>/bench_trace.json                 the JSON line bench.py printed under the trace pass

The bench's last `isolated` launches (one per HIP-event reading, untimed pass at the end of the
headline measurement) are the unit: with batches >= 32768 queries a launch is two dispatches of
ph_search_kernel (upper layers, then the bottom layer in locality order)."""
import csv
import glob
import json
import os
import sys


def find(d, pat):
    r = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))
    if not r:
        raise SystemExit("missing %s under %s" % (pat, d))
    return r[-1]


def search_rows(path, name_col="Kernel_Name"):
    rows = list(csv.DictReader(open(path)))
    return [r for r in rows if "ph_search_kernel" in r[name_col]]


def main():
    d, out_dir = sys.argv[1], sys.argv[2]
    isolated = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    bench = json.loads(open(os.path.join(d, "bench_trace.json")).read().strip().splitlines()[-1])
    per = bench["roofline"].get("dispatches_per_launch", 1)
    tr = search_rows(find(os.path.join(d, "trace"), "*kernel_trace.csv"))
    tr.sort(key=lambda r: int(r["Start_Timestamp"]))
    last = tr[-isolated * per:]
    launches = [last[i * per:(i + 1) * per] for i in range(isolated)]
    dur = [sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in L) / 1e6 for L in launches]
    span = [(int(L[-1]["End_Timestamp"]) - int(L[0]["Start_Timestamp"])) / 1e6 for L in launches]
    per_dispatch = [[(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in L] for L in launches]
    summary = {
        "command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py " + bench.get("argv", ""),
        "kernel": last[-1]["Kernel_Name"],
        "isolated_launches": isolated,
        "dispatches_per_launch": per,
        "avg_ms_sum_of_dispatches": sum(dur) / len(dur),
        "avg_ms_first_start_to_last_end": sum(span) / len(span),
        "avg_ms_per_dispatch": [sum(x[i] for x in per_dispatch) / len(per_dispatch) for i in range(per)],
        "min_ms": min(dur), "max_ms": max(dur),
        "bench_reported_kernel_ms": bench["roofline"]["kernel_ms"],
        "bench_roofline": bench["roofline"],
        "bench_value": bench["value"], "bench_config": bench["config"],
    }
    pm = {}
    for key, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        try:
            rows = search_rows(find(os.path.join(d, sub), "*counter_collection.csv"))
        except SystemExit:
            continue
        rows = [r for r in rows if r["Counter_Name"] == key]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        iso_p = 5  # the PMC passes run bench.py with --steps 5: five isolated launches close the run
        lastp = rows[-iso_p * per:]
        vals = [sum(float(r["Counter_Value"]) for r in lastp[i * per:(i + 1) * per]) for i in range(iso_p)]
        pm[key + "_KB_per_launch"] = sum(vals) / len(vals)
    if "FETCH_SIZE_KB_per_launch" in pm:
        fetch = pm["FETCH_SIZE_KB_per_launch"] * 1024 * 2  # gfx950 correction, see below
        write = pm.get("WRITE_SIZE_KB_per_launch", 0.0) * 1024
        pm["correction"] = ("gfx950: FETCH_SIZE counts 64 B per 128-B request for 16 B/lane coalesced loads -> doubled "
                            "(MI355X_MICROARCH.md section HBM); WRITE_SIZE taken as is")
        pm["traffic_bytes_per_launch"] = fetch + write
        pm["command"] = ("rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv -- "
                         "python3 bench.py (same flags, --steps 5 --warmup 1)")
        w = bench["config"]
        pm["workload"] = {"dataset": "clustered", "n": 1000000, "dim": 768, "nq": bench.get("queries_per_step_per_gpu"),
                          "ef": w["number_of_candidates"], "upper": w["upper_layer_candidate_count"],
                          "probe_depth": w["probe_depth"]}
        pm["algorithmic_bytes_per_launch"] = bench["roofline"]["algorithmic_bytes_per_launch"]
    summary["pmc"] = pm
    os.makedirs(out_dir, exist_ok=True)
    json.dump(summary, open(os.path.join(out_dir, "search_kernel_summary.json"), "w"), indent=1)
    # the rocprofv3 --stats table itself
    try:
        st = find(os.path.join(d, "trace"), "*kernel_stats.csv")
        open(os.path.join(out_dir, "kernel_stats_bench.csv"), "w").write(open(st).read())
    except SystemExit:
        pass
    print(json.dumps(summary, indent=1)[:3000])


if __name__ == "__main__":
    main()
