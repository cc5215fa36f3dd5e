"""Copy what scripts/round_records.sh left under gpurun_out/<dir> into profiles/ under the round's names, and print the figures
profiles/README.md quotes (the README itself is edited by hand).
    python scripts/install_records.py gpurun_out/r4_rec7 r04"""
import csv
import glob
import json
import os
import shutil
import sys

R, tag = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "r04")
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def last(path):
    return json.loads([l for l in open(path) if l.startswith("{")][-1])


for f in glob.glob(os.path.join(R, "profiles", "*")):
    shutil.copy(f, P)
pairs = {"bench_driver_cmd.json": f"{tag}_bench.json", "bench_under_rocprof.json": f"{tag}_bench_under_rocprof.json",
         "size_scaling.jsonl": f"{tag}_size_scaling.jsonl", "multirank_rehearsals.jsonl": f"{tag}_multirank_rehearsals.jsonl",
         "persist_phases_blocks.json": f"{tag}_persist_phases_blocks.json",
         "persist_phases_triangles.json": f"{tag}_persist_phases_triangles.json",
         "persist_phases_mg.json": f"{tag}_persist_phases_mg.json", "pmc_onchip_compare.json": f"{tag}_pmc_onchip_compare.json"}
for src, dst in pairs.items():
    shutil.copy(os.path.join(R, src), os.path.join(P, dst))
shutil.copy(glob.glob(os.path.join(R, "stats", "*", "*_kernel_stats.csv"))[0], os.path.join(P, f"{tag}_kernel_stats.csv"))

print("pmc", json.load(open(os.path.join(P, "pmc_summary.json")))["_meta"])
for f in ("bench.json", "bench_driver_cmd.json", "bench_under_rocprof.json"):
    d = last(os.path.join(R, f))
    print(f, "value %.2f M  %.2f ms/step  %.3f us/it  frac %.4f  stale %s | unstructured %.3f us %.2f M ok %s | assembly %.4f ms frac %.3f | "
          "16M spmv %.1f us %.3f, iteration %.1f us %.3f" %
          (d["value"] / 1e6, d["ms_per_step"], d["roofline"]["us_per_iteration"], d["roofline"]["frac"], d["roofline"].get("traffic_stale"),
           d["unstructured"]["us_per_iteration"], d["unstructured"]["value"] / 1e6, d["unstructured"]["fixture_parity"]["ok"],
           d["assembly"]["ms"], d["assembly"]["frac"], d["hbm_resident"]["spmv"]["us_per_launch"], d["hbm_resident"]["spmv"]["frac"],
           d["hbm_resident"]["iteration"]["us_per_launch"], d["hbm_resident"]["iteration"]["frac"]))
    print("   phases", {k: round(v, 4) for k, v in d["phases_ms"].items()})
print("HIP events under rocprof: us per launch", last(os.path.join(R, "bench_under_rocprof.json"))["roofline"]["us_per_launch"])
for l in open(os.path.join(R, "size_scaling.jsonl")):
    if l.startswith("{"):
        d = json.loads(l)
        print("size", d["config"]["workload"][:14], d["dtype"], "%.2f M %.2f ms frac %.3f %.3f us mode %s k %s it %d" %
              (d["value"] / 1e6, d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("us_per_iteration") or 0,
               d["roofline"].get("edge_block_mode"), d["roofline"].get("tiles_per_workgroup"), d["cg_iterations"]))
for l in open(os.path.join(R, "multirank_rehearsals.jsonl")):
    if l.startswith("{"):
        d = json.loads(l)
        print("ranks", d["config"]["workload"][:12], d["n_gpus"], "%.2f ms %.3f us mode %s it %d parity %s ms_order %.2f" %
              (d["ms_per_step"], d["roofline"].get("us_per_iteration") or 0, d["roofline"].get("edge_block_mode"), d["cg_iterations"],
               d["fixture_parity"]["ok_on_every_rank"], d["phases_ms"]["ms_order"]))
for r in csv.DictReader(open(os.path.join(P, f"{tag}_kernel_stats.csv"))):
    if any(k in r["Name"] for k in ("k_cg_persist", "k_assemble_fan", "k_ring16", "k_rhs_touched", "k_cg_fused_dma", "k_count_degree")):
        print("stats", r["Name"][:72], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
d = json.load(open(os.path.join(P, "pmc_summary.json")))["hole1m:tile512:kernel2"]
print("pmc on-chip", d["kernel"], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["utilisation"].items()})
for r in json.load(open(os.path.join(R, "persist_phases_blocks.json")))["runs"]:
    print("phases", r["workload"], r.get("node_slots_per_lane"), round(r["us_per_iteration_product_build"], 3),
          round(r["us_per_iteration_stamped_build"], 3), {k: round(v, 2) for k, v in r["phases_us_mean_over_workgroups"].items()},
          {k: round(v, 2) for k, v in r["detail_us_mean"].items()})
for k, v in json.load(open(os.path.join(R, "pmc_onchip_compare.json"))).items():
    print("compare", k, round(v["cycles_per_iteration"]), v["SQ_INSTS_VALU"], v["SQ_WAIT_ANY"], round(v["lds_conflict_share"], 3))
