"""Turns the raw rocprofv3 output of profiles/collect.sh (gpurun_out/<tag>/) into the committed summaries:
   profiles/<round>_kernel_stats.csv, <round>_kernel_stats_graph.csv, <round>_pmc.csv, <round>_traffic.json
usage: python profiles/summarise.py <tag> <round-prefix, e.g. r01>"""
import collections
import csv
import glob
import json
import re
import sys

tag, rnd = sys.argv[1], sys.argv[2]
config = sys.argv[3] if len(sys.argv) > 3 else "cfg3"
root = f"gpurun_out/{tag}"

# the bench's one-off parity call runs other template instances than the timed per-view path: keep them apart
SHORT = [("k_render_bwd3<false, false, true", "render_bwd_colour_grad"), ("k_render_bwd3<true, false, true", "render_bwd_colour_allmap"),
         ("k_render_bwd3<true, true, true", "render_bwd_all_grad"), ("k_render_bwd3<false, false, false", "render_bwd_training_general"),
         ("k_render_bwd_unit<16", "render_bwd_unit_gated"),
         ("k_render_fwd3<true, false", "render_fwd_presorted"), ("k_render_fwd3<true, true, false", "render_fwd_general"),
         ("k_render_bwd", "render_bwd"), ("k_render_fwd", "render_fwd"), ("k_tile_rank_sort", "tile_sort"),
         ("k_tile_sort", "tile_sort_big"), ("k_scatter", "scatter"), ("k_preprocess_fwd", "preprocess_fwd"),
         ("k_preprocess_bwd", "preprocess_bwd"), ("k_scan_tiles", "scan_tiles"), ("k_sample_f12", "sample_f12"),
         ("k_sample_f3", "sample_f3"), ("k_sample_bwd<1>", "sample_b1"),
          ("k_sample_bwd<3>", "sample_b3"), ("k_sample_bwd_close", "sample_b3"), ("k_attrs_fwd", "attrs_fwd"),
         ("k_attrs_bwd", "attrs_bwd"), ("k_zero_vec", "zero_fill"), ("k_zero_words", "zero_fill"),
         ("k_view_fwd", "view_fwd"), ("k_view_bwd", "view_bwd")]


def short(name):
    for pat, s in SHORT:
        if pat in name:
            return s
    return None


def stats(sub, out, header):
    f = sorted(glob.glob(f"{root}/{sub}/*/*_kernel_stats.csv"))[-1]
    rows = list(csv.DictReader(open(f)))
    with open(out, "w") as o:
        o.write(header)
        o.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        for r in rows:
            nm = re.sub(r"\(.*", "", r["Name"])
            if float(r["Percentage"]) < 0.05:
                continue
            o.write(f'"{nm}",{r["Calls"]},{r["TotalDurationNs"]},{float(r["AverageNs"]):.1f},{r["Percentage"]},{r["MinNs"]},{r["MaxNs"]}\n')


stats("trace", f"profiles/{rnd}_kernel_stats.csv",
      "# rocprofv3 --kernel-trace --stats -- python bench.py --streams 1 --views-per-step 1 --no-graph --no-cpu-baseline "
      "--no-kernel-times --no-train-step --steps 16 --warmup 2   (cfg3, serial eager schedule; MI355X gfx950)\n")
stats("trace_graph", f"profiles/{rnd}_kernel_stats_graph.csv",
      "# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-kernel-times --no-train-step --steps 16 "
      "--warmup 2   (default schedule: 3 views in flight, one hipGraph replay per view; kernels of different views overlap, "
      "so per-kernel durations are inflated relative to the serial schedule)\n")

if glob.glob(f"{root}/trace_general/*/*_kernel_stats.csv"):
    stats("trace_general", f"profiles/{rnd}_kernel_stats_general.csv",
          "# rocprofv3 --kernel-trace --stats -- python profiles/probes/general_instances.py cfg3 12   (the operator API, "
          "GaussianRasterizer: every backward instance -- k_render_bwd_unit<16, true> gated unit kernel of the reference's own call, "
          "k_render_bwd3<false, false, false> training instance with the unit route off, <false, false, true> colour gradient, "
          "<true, false, true> + all_map, <true, true, true> + inverse depth --, the general sorting forward "
          "k_render_fwd3<true, true, false, true>; cfg3 view 0; MI355X gfx950)\n")

agg = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
# one counter file per pass: gpurun MERGES a new collection into an existing gpurun_out/<tag>/, so a pass directory can hold
# the files of an earlier collection as well -- only the newest of each pass counts
import os
newest = {}
for f in glob.glob(f"{root}/*/*/*_counter_collection.csv"):
    k = f.split(os.sep)[-3]
    if k not in newest or os.path.getmtime(f) > os.path.getmtime(newest[k]):
        newest[k] = f
for f in newest.values():
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k:
            agg[k][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
with open(f"profiles/{rnd}_pmc.csv", "w") as o:
    o.write("# rocprofv3 --pmc <8 counters> --kernel-trace, three SQ passes + FETCH_SIZE + WRITE_SIZE passes (profiles/collect.sh); "
            "per-launch averages (summed over XCDs/SEs), serial eager schedule, cfg3\nKernel,Counter,AvgPerLaunch\n")
    for k in sorted(agg):
        for c in sorted(agg[k]):
            v = agg[k][c]
            o.write(f"{k},{c},{sum(v.values()) / len(v):.1f}\n")
traffic = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, profiles/collect.sh), bench.py --steps 4 serial "
                   "eager view mode cfg3, per-launch averages; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE "
                   "half-count correction of MI355X_MICROARCH.md; WRITE_SIZE uncalibrated)",
           "round": int(re.match(r"r(\d+)", rnd).group(1)), "snapshot": tag, "config": config, "kernels": {}}
# duration of each kernel in the trace pass of the same collection (same box, same command): the scale the counters belong to
trace_us = collections.defaultdict(lambda: [0.0, 0])
tf = sorted(glob.glob(f"{root}/trace/*/*_kernel_stats.csv"))
if tf:
    for r in csv.DictReader(open(tf[-1])):
        k = short(r["Name"])
        if k:
            trace_us[k][0] += float(r["TotalDurationNs"]) * 1e-3
            trace_us[k][1] += int(r["Calls"])
for k in agg:
    if "FETCH_SIZE" in agg[k] and "WRITE_SIZE" in agg[k]:
        fv, wv = agg[k]["FETCH_SIZE"], agg[k]["WRITE_SIZE"]
        fkb, wkb = sum(fv.values()) / len(fv), sum(wv.values()) / len(wv)
        traffic["kernels"][k] = {"FETCH_SIZE_KB": round(fkb, 1), "WRITE_SIZE_KB": round(wkb, 1),
                                 "hbm_bytes_per_launch": int((2 * fkb + wkb) * 1024)}
        if trace_us[k][1]:
            traffic["kernels"][k]["kernel_us_in_trace"] = round(trace_us[k][0] / trace_us[k][1], 2)
json.dump(traffic, open(f"profiles/{rnd}_traffic.json", "w"), indent=1)
print("wrote", rnd, "kernels with traffic:", sorted(traffic["kernels"]))
