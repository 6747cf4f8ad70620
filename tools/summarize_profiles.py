#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of tools/profile_round.sh into the files committed under profiles/.

    python tools/summarize_profiles.py gpurun_out/prof_r01 r01

Writes profiles/<tag>_kernel_stats_default_cmd.csv, <tag>_kernel_stats_depth1.csv, <tag>_pmc_fetch_write.csv,
<tag>_pmc_valu.csv and <tag>_msm_traffic.json (read back by bench.py for roofline.traffic)."""
import csv
import json
import os
import sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)


def short(name):
    return name.split("(")[0].replace("void ", "")


def stats(sub, dst, header):
    rows = list(csv.DictReader(open(os.path.join(src, sub, "k_kernel_stats.csv"))))
    with open(os.path.join(out, dst), "w") as f:
        for h in header:
            f.write("# " + h + "\n")
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
        for r in rows:
            f.write('"%s",%s,%s,%s,%s\n' % (r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
    return {short(r["Name"]): float(r["AverageNs"]) for r in rows}


def bench_line(name):
    for line in open(os.path.join(src, name)):
        if line.startswith("{"):
            return json.loads(line)
    return {}


bd, b1 = bench_line("default.json"), bench_line("depth1.json")
stats("default", f"{tag}_kernel_stats_default_cmd.csv",
      ["rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline   (the default bench command; MI355X)",
       "%s" % bd.get("config", {}).get("workload", ""),
       "bench line of this run: value=%.0f proofs/s, ms_per_step=%.4f (kernels of different launches overlap, durations are inflated by sharing)" % (bd.get("value", 0), bd.get("ms_per_step", 0))])
avg1 = stats("depth1", f"{tag}_kernel_stats_depth1.csv",
             ["rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 128 --warmup 32 --depth 1 --no-cpu-baseline",
              "one launch in flight (32 steps = 32 x 1024 proofs per launch), so per-kernel durations are undisturbed",
              "bench line of this run: value=%.0f proofs/s, stages_ms=%s" % (b1.get("value", 0), json.dumps(b1.get("stages_ms", {})))])


def counters(sub):
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(os.path.join(src, sub, "k_counter_collection.csv"))):
        k = (short(r["Kernel_Name"]), r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


fetch, write, valu = counters("pmc_fetch"), counters("pmc_write"), counters("pmc_valu")
with open(os.path.join(out, f"{tag}_pmc_fetch_write.csv"), "w") as f:
    f.write("# rocprofv3 --pmc FETCH_SIZE and (separate pass) --pmc WRITE_SIZE -- python3 bench.py --steps 128 --warmup 32 --depth 1 --no-cpu-baseline\n")
    f.write("# mean counter value per dispatch; unit KB (bytes = value * 1024); on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM)\n")
    f.write("Kernel,Counter,Dispatches,MeanValueKB\n")
    for d in (fetch, write):
        for (k, c), (v, n) in sorted(d.items(), key=lambda kv: -kv[1][0]):
            if k.startswith("h2v::"):
                f.write('"%s",%s,%d,%.1f\n' % (k, c, n, v))
with open(os.path.join(out, f"{tag}_pmc_valu.csv"), "w") as f:
    f.write("# rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY -- python3 bench.py --steps 128 --warmup 32 --depth 1 --no-cpu-baseline\n")
    f.write("# mean per dispatch; SQ_* cycle counters are in quad-cycles; valu_active = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (share of wave time spent issuing VALU)\n")
    f.write("Kernel,Dispatches,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_ACTIVE_INST_VALU,SQ_INSTS_VALU,SQ_WAIT_INST_ANY,valu_active\n")
    ks = sorted({k for (k, c) in valu if k.startswith("h2v::")}, key=lambda k: -valu.get((k, "SQ_WAVE_CYCLES"), (0, 0))[0])
    for k in ks:
        g = lambda c: valu.get((k, c), (0.0, 0))[0]
        wc = g("SQ_WAVE_CYCLES")
        f.write('"%s",%d,%.0f,%.0f,%.0f,%.0f,%.0f,%.3f\n' % (k, valu[(k, "SQ_WAVE_CYCLES")][1], wc, g("SQ_BUSY_CYCLES"), g("SQ_ACTIVE_INST_VALU"), g("SQ_INSTS_VALU"),
                                                            g("SQ_WAIT_INST_ANY"), g("SQ_ACTIVE_INST_VALU") / wc if wc else 0))

msm = sorted({k for (k, c) in list(fetch) + list(write) if k.startswith("h2v::msm_")})
per = {k: {"FETCH_SIZE": fetch.get((k, "FETCH_SIZE"), (0, 0))[0], "WRITE_SIZE": write.get((k, "WRITE_SIZE"), (0, 0))[0]} for k in msm}
fk, wk = sum(v["FETCH_SIZE"] for v in per.values()), sum(v["WRITE_SIZE"] for v in per.values())
terms = b1.get("roofline", {}).get("terms_per_launch")
json.dump({
    "source": f"profiles/{tag}_pmc_fetch_write.csv (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; bench.py --depth 1, %s steps per launch)" % b1.get("config", {}).get("steps_per_launch"),
    "correction": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE reads half of wide coalesced fetches on gfx950 (MI355X_MICROARCH.md, HBM)",
    "fetch_kb": fk, "write_kb": wk,
    "msm_stage_traffic_bytes_per_launch": (2 * fk + wk) * 1024,
    "terms_per_launch": terms,
    "algorithmic_bytes_per_launch": 96 * terms if terms else None,
    "msm_kernel_avg_ns_depth1": {k: avg1.get(k) for k in msm},
    "per_kernel_kb": per,
}, open(os.path.join(out, f"{tag}_msm_traffic.json"), "w"), indent=1)
print(open(os.path.join(out, f"{tag}_msm_traffic.json")).read())
print(open(os.path.join(out, f"{tag}_pmc_valu.csv")).read())
print(open(os.path.join(out, f"{tag}_kernel_stats_depth1.csv")).read())
