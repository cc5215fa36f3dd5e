import json, os, sys
sys.path.insert(0, os.getcwd())
from magnetite_amd import Context, meshgen
for n in (29, 36, 40, 44):
    prob = meshgen.config_fixed_left_pull_right(meshgen.plate(n))
    row = {"nodes": prob.mesh.num_nodes}
    for mode in ("single", "grid"):
        if mode == "grid": os.environ["MAG_TUNE_PERSIST_SINGLE_WG"] = "0"
        else: os.environ.pop("MAG_TUNE_PERSIST_SINGLE_WG", None)
        with Context(device=0) as c:
            c.upload_problem(prob); c.run(); best = 1e9
            for _ in range(5):
                c.run(); st = c.stats(); best = min(best, st["ms_cg"] * 1e3 / st["iterations"])
        row[mode] = {"k": st["tiles_per_workgroup"], "us": round(best, 3), "it": st["iterations"]}
    print(json.dumps(row), flush=True)
