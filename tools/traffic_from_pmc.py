"""HBM bytes per path of the bench workloads from the counter summaries tools/pmc_summary.py wrote:
python tools/traffic_from_pmc.py DIR_WITH_<wl>_pmc_summary.json OUT.json
One launch of tools/profile_target.py is 1920x1080 x 32 spp.  WRITE_SIZE is taken as it reads,
FETCH_SIZE doubled (its gfx950 correction); both count what passed the L2's memory side."""
import json, os, sys
src, out = sys.argv[1], sys.argv[2]
paths = 1920 * 1080 * 32
res = {}
for wl in ("cfg2", "cfg3", "cfg4", "cfg5"):
    p = os.path.join(src, f"{wl}_pmc_summary.json")
    if not os.path.exists(p):
        continue
    d = json.load(open(p))["derived"]
    w, f = d.get("hbm_write_bytes_per_launch", 0.0), d.get("hbm_fetch_bytes_per_launch_corrected_x2", 0.0)
    res[wl] = {"hbm_bytes_per_path": (w + f) / paths, "write_bytes_per_launch": w, "fetch_bytes_per_launch": f, "paths_per_launch": paths,
               "source": f"{p}: rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE in separate passes, one launch of 1920x1080 x 32 spp"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
