#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of tools/profile_round.sh into the files committed under profiles/.

    python tools/summarize_profiles.py gpurun_out/prof_r02 r02

Writes profiles/<tag>_kernel_stats_driver_cmd.csv, <tag>_kernel_stats_default_cmd.csv, and for the two launch shapes
(driver: 20 steps per launch; d1: 32 steps per launch, one launch in flight) <tag>_pmc_fetch_write_<shape>.csv,
<tag>_pmc_valu_<shape>.csv and <tag>_msm_traffic_<shape>.json (read back by bench.py for roofline.traffic and
roofline.kernels.*.valu_active, picked by launch shape)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]

# FETCH_SIZE -> bytes, per kernel, from the calibration of the access patterns (profiles/r03_fetch_calibration.txt): wide (whole-line)
# reads are counted at half their size on gfx950, sub-line gathers in full.  WRITE_SIZE is exact for every pattern measured.
FETCH_FACTOR_DEFAULT = 2.0
FETCH_FACTOR = {"h2v::msm_accumulate": (1.0, "72-B base gathers: rows k_gather72 / k_gather72_big count the straddled sectors in full; its streamed list reads (4 B per entry) are < 4 % of its fetches")}
FETCH_REASON_DEFAULT = "coalesced streams / whole 128-byte slots: rows k_stream_read, k_stream4, k_gather128 count half"


def fetch_factor(kernel):
    for k, (f, why) in FETCH_FACTOR.items():
        if kernel.startswith(k):
            return f, why
    return FETCH_FACTOR_DEFAULT, FETCH_REASON_DEFAULT
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)


def short(name):
    return name.split("(")[0].replace("void ", "")


def find(sub, leaf):
    hits = glob.glob(os.path.join(src, sub, "**", leaf), recursive=True)
    return hits[0] if hits else None


def bench_line(name):
    p = os.path.join(src, name)
    if os.path.exists(p):
        for line in open(p):
            if line.startswith("{"):
                return json.loads(line)
    return {}


def stats(sub, dst, header):
    f = find(sub, "k_kernel_stats.csv")
    if not f:
        return {}
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(out, dst), "w") as o:
        for h in header:
            o.write("# " + h + "\n")
        o.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
        for r in rows:
            o.write('"%s",%s,%s,%s,%s\n' % (r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
    return {short(r["Name"]): float(r["AverageNs"]) for r in rows}


def counters(sub):
    f = find(sub, "k_counter_collection.csv")
    acc = defaultdict(lambda: [0.0, 0])
    if f:
        for r in csv.DictReader(open(f)):
            k = (short(r["Kernel_Name"]), r["Counter_Name"])
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def headline_stats(sub, dst, header):
    """Per-kernel averages over the HEADLINE launches of the driver command only.  The command also runs the PCIe-inclusive leg (same
    launch shape: kept), the config-4 leg (another VK, launches in flight) and the SingleStrategy leg (512 one-proof groups) under the
    same kernel names; a launch is the run of dispatches from one k_decompress to the next, and it counts when its k_decompress has the
    grid of the very first one (the warm-up launch of the headline)."""
    f = find(sub, "k_kernel_trace.csv")
    if not f:
        return {}
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    gkey = next(k for k in ("Grid_Size_X", "Grid_Size", "Workgroup_Count_X") if k in rows[0])
    dec = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]).endswith("k_decompress")]
    if not dec:
        return {}
    want = rows[dec[0]][gkey]
    acc = defaultdict(lambda: [0.0, 0])
    launches = 0
    for a, b in zip(dec, dec[1:] + [len(rows)]):
        if rows[a][gkey] != want:
            continue
        launches += 1
        for r in rows[a:b]:
            k = r["Kernel_Name"]
            acc[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); acc[k][1] += 1
    with open(os.path.join(out, dst), "w") as o:
        for h in header + [f"averages over the {launches} launches of the headline shape (k_decompress grid {want}); the all-legs table of rocprofv3 --stats is in {dst.replace('.csv', '_all_legs.csv')}"]:
            o.write("# " + h + "\n")
        o.write("Name,Calls,TotalDurationNs,AverageNs\n")
        for k, (t, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
            o.write('"%s",%d,%d,%.1f\n' % (k, n, t, t / n))
    return {short(k): t / n for k, (t, n) in acc.items()}


bdrv, bdef = bench_line("driver.json"), bench_line("default.json")
avg_all = stats("driver", f"{tag}_kernel_stats_driver_cmd_all_legs.csv",
                ["rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5   (the command the driver runs; MI355X)",
                 "%s" % bdrv.get("config", {}).get("workload", ""),
                 "calls: 1 warm-up + 7 timed launches (the timed region repeated, median reported) + 4 launches re-timed alone after the timed region, the same again for the PCIe-inclusive leg, + the config-4 and SingleStrategy legs; every headline launch carries 20 steps",
                 "bench line of this run: value=%.0f proofs/s, ms_per_step=%.4f, stages_ms_one_launch_in_flight=%s" % (bdrv.get("value", 0), bdrv.get("ms_per_step", 0), json.dumps(bdrv.get("stages_ms_one_launch_in_flight", {})))])
avg_drv = headline_stats("driver", f"{tag}_kernel_stats_driver_cmd.csv",
                         ["rocprofv3 --kernel-trace -- python3 bench.py --gpus 1 --steps 20 --warmup 5   (the command the driver runs; MI355X)",
                          "%s" % bdrv.get("config", {}).get("workload", ""),
                          "bench line of this run: value=%.0f proofs/s, ms_per_step=%.4f, stages_ms_one_launch_in_flight=%s" % (bdrv.get("value", 0), bdrv.get("ms_per_step", 0), json.dumps(bdrv.get("stages_ms_one_launch_in_flight", {})))]) or avg_all
stats("default", f"{tag}_kernel_stats_default_cmd.csv",
      ["rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline   (the default bench command: 32 steps per launch, 8 launches in flight; MI355X)",
       "%s" % bdef.get("config", {}).get("workload", ""),
       "bench line of this run: value=%.0f proofs/s, ms_per_step=%.4f (kernels of different launches overlap, durations are inflated by sharing)" % (bdef.get("value", 0), bdef.get("ms_per_step", 0))])

for shape, pre, cmd in (("steps20", "drv", "python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-reupload-leg"),
                        ("steps32", "d1", "python3 bench.py --steps 128 --warmup 32 --depth 1 --no-cpu-baseline --no-reupload-leg")):
    fetch, write, valu = counters(pre + "_fetch"), counters(pre + "_write"), counters(pre + "_valu")
    if not fetch and not write:
        continue
    b1 = bench_line(pre + "_fetch.json")
    with open(os.path.join(out, f"{tag}_pmc_fetch_write_{shape}.csv"), "w") as f:
        f.write(f"# rocprofv3 --pmc FETCH_SIZE and (separate pass) --pmc WRITE_SIZE -- {cmd}\n")
        f.write("# mean counter value per dispatch; unit KB (bytes = value * 1024); on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM)\n")
        f.write("Kernel,Counter,Dispatches,MeanValueKB\n")
        for d in (fetch, write):
            for (k, c), (v, n) in sorted(d.items(), key=lambda kv: -kv[1][0]):
                if k.startswith("h2v::"):
                    f.write('"%s",%s,%d,%.1f\n' % (k, c, n, v))
    valu_active = {}
    with open(os.path.join(out, f"{tag}_pmc_valu_{shape}.csv"), "w") as f:
        f.write(f"# rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY -- {cmd}\n")
        f.write("# mean per dispatch; SQ_* cycle counters are in quad-cycles; valu_active = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (share of wave time spent issuing VALU)\n")
        f.write("Kernel,Dispatches,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_ACTIVE_INST_VALU,SQ_INSTS_VALU,SQ_WAIT_INST_ANY,valu_active\n")
        ks = sorted({k for (k, c) in valu if k.startswith("h2v::")}, key=lambda k: -valu.get((k, "SQ_WAVE_CYCLES"), (0, 0))[0])
        for k in ks:
            g = lambda c: valu.get((k, c), (0.0, 0))[0]
            wc = g("SQ_WAVE_CYCLES")
            valu_active[k] = g("SQ_ACTIVE_INST_VALU") / wc if wc else 0
            f.write('"%s",%d,%.0f,%.0f,%.0f,%.0f,%.0f,%.3f\n' % (k, valu[(k, "SQ_WAVE_CYCLES")][1], wc, g("SQ_BUSY_CYCLES"), g("SQ_ACTIVE_INST_VALU"), g("SQ_INSTS_VALU"),
                                                                g("SQ_WAIT_INST_ANY"), valu_active[k]))
    msm = sorted({k for (k, c) in list(fetch) + list(write) if k.startswith("h2v::msm_")})
    per = {k: {"FETCH_SIZE": fetch.get((k, "FETCH_SIZE"), (0, 0))[0], "WRITE_SIZE": write.get((k, "WRITE_SIZE"), (0, 0))[0]} for k in msm}
    allk = sorted({k for (k, c) in list(fetch) + list(write) if k.startswith("h2v::")})
    per_all = {k: {"FETCH_SIZE": fetch.get((k, "FETCH_SIZE"), (0, 0))[0], "WRITE_SIZE": write.get((k, "WRITE_SIZE"), (0, 0))[0]} for k in allk}
    fk, wk = sum(v["FETCH_SIZE"] for v in per.values()), sum(v["WRITE_SIZE"] for v in per.values())
    terms = b1.get("roofline", {}).get("terms_per_launch")
    per_bytes = {k: (fetch_factor(k)[0] * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 for k, v in per.items()}
    json.dump({
        "source": f"profiles/{tag}_pmc_fetch_write_{shape}.csv (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; {cmd}; %s steps per launch)" % b1.get("config", {}).get("steps_per_launch"),
        "correction": {"rule": "bytes = (fetch_factor * FETCH_SIZE + WRITE_SIZE) * 1024, fetch_factor per kernel from the calibrated access patterns (profiles/r03_fetch_calibration.txt)",
                       "per_kernel": {k: {"fetch_factor": fetch_factor(k)[0], "calibration": fetch_factor(k)[1]} for k in msm},
                       "round2": "round 2 doubled FETCH_SIZE for every kernel: %.0f bytes for this launch" % ((2 * fk + wk) * 1024)},
        "fetch_kb": fk, "write_kb": wk,
        "msm_stage_traffic_bytes_per_launch": sum(per_bytes.values()),
        "per_kernel_bytes": per_bytes,
        "terms_per_launch": terms,
        "algorithmic_bytes_per_launch": 96 * terms if terms else None,
        "msm_kernel_avg_ns": {k: avg_drv.get(k) for k in msm} if shape == "steps20" else None,
        "per_kernel_kb": per,
        "all_kernels_kb": per_all,
        "valu_active": valu_active,
    }, open(os.path.join(out, f"{tag}_msm_traffic_{shape}.json"), "w"), indent=1)
    print(open(os.path.join(out, f"{tag}_msm_traffic_{shape}.json")).read()[:1500])
for f in sorted(glob.glob(os.path.join(out, f"{tag}_kernel_stats_driver_cmd.csv"))):
    print(open(f).read())
