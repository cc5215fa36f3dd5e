"""Kernel time of the ordering phase + CSR pattern per rank and solve, from rocprofv3 kernel stats of
scripts/order_phase_ranks.py (one mode per trace; eight ranks x three solves in the trace).
    python scripts/order_phase_ranks_summarize.py <kernel_stats.csv> [ranks=8] [solves=3]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
solves = int(sys.argv[3]) if len(sys.argv) > 3 else 3
CG = ("k_cg_", "k_operator", "k_fused", "k_stream", "k_element_stress", "k_scatter_back", "k_reactions", "k_assemble", "k_rhs",
      "copyBuffer", "k_gather", "k_to_hilbert", "k_from_hilbert")
tot, lines = 0.0, []
for r in rows:
    if any(c in r["Name"] for c in CG):
        continue
    us = float(r["TotalDurationNs"]) / 1e3 / (ranks * solves)
    tot += us
    lines.append((us, int(r["Calls"]) // (ranks * solves), r["Name"][:90]))
lines.sort(reverse=True)
print(f"symbolic kernels (ordering phase + CSR pattern, fills included): {tot / 1e3:.3f} ms per rank and solve")
for us, calls, name in lines[:16]:
    print(f"  {us:9.1f} us  x{calls:<3d} {name}")
