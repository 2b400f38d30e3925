"""HBM bytes per path of the bench workloads from the counter summaries tools/pmc_summary.py wrote:
python tools/traffic_from_pmc.py DIR_WITH_<name>_pmc_summary.json OUT.json
Each summary names the launch its counters belong to (1920x1080 x 32 spp unless tools/profile.sh was
given another shape, e.g. cfg5_8k: the bench's own 7680x4320 x 16 spp).  WRITE_SIZE is taken as it
reads, FETCH_SIZE doubled (its gfx950 correction); both count what passed the L2's memory side."""
import glob, json, os, sys
src, out = sys.argv[1], sys.argv[2]
res = {}
for p in sorted(glob.glob(os.path.join(src, "*_pmc_summary.json"))):
    name = os.path.basename(p)[:-len("_pmc_summary.json")]
    doc = json.load(open(p))
    d, launch = doc["derived"], doc.get("launch") or {"width": 1920, "height": 1080, "spp": 32}
    paths = launch["width"] * launch["height"] * launch["spp"]
    w, f = d.get("hbm_write_bytes_per_launch", 0.0), d.get("hbm_fetch_bytes_per_launch_corrected_x2", 0.0)
    res[name] = {"hbm_bytes_per_path": (w + f) / paths, "write_bytes_per_path": w / paths, "fetch_bytes_per_path": f / paths,
                 "write_bytes_per_launch": w, "fetch_bytes_per_launch": f, "paths_per_launch": paths,
                 "launch": f"{launch['width']}x{launch['height']} x {launch['spp']} spp",
                 "source": f"{p}: rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE in separate passes, one launch of {launch['width']}x{launch['height']} x {launch['spp']} spp"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
