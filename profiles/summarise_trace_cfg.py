"""usage: python profiles/summarise_trace_cfg.py <tag> <round-prefix> <cfgN> [...]  ->  profiles/<round>_kernel_stats_<cfg>.csv
(the per-config companions of <round>_kernel_stats.csv: serial eager schedule of bench.py's headline body, profiles/collect_trace_cfg.sh)"""
import csv
import glob
import re
import sys

tag, rnd, cfgs = sys.argv[1], sys.argv[2], sys.argv[3:]
for cfg in cfgs:
    f = sorted(glob.glob(f"gpurun_out/{tag}/trace_{cfg}/*/*_kernel_stats.csv"))[-1]
    rows = list(csv.DictReader(open(f)))
    with open(f"profiles/{rnd}_kernel_stats_{cfg}.csv", "w") as o:
        o.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py --config {cfg} --streams 1 --views-per-step 1 --no-graph "
                "--no-cpu-baseline --no-kernel-times --no-train-step --min-seconds 0 --steps 16 --warmup 2   (serial eager schedule; "
                "MI355X gfx950; the k_render_bwd3 / k_preprocess_* / general k_render_fwd3 rows come from bench.py's per-camera capacity "
                "pre-pass through the operator API, not from the timed per-view body)\n")
        o.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        for r in rows:
            if float(r["Percentage"]) < 0.05:
                continue
            nm = re.sub(r"\(.*", "", r["Name"])
            o.write(f'"{nm}",{r["Calls"]},{r["TotalDurationNs"]},{float(r["AverageNs"]):.1f},{r["Percentage"]},{r["MinNs"]},{r["MaxNs"]}\n')
    print("wrote", cfg)
