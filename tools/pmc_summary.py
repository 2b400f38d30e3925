"""Summarises the rocprofv3 counter passes that tools/profile.sh wrote: sums every counter over
the dispatches of the render kernel and derives the ratios quoted in the design notes.
python tools/pmc_summary.py gpurun_out/prof_<tag>_<workload> [out.json]"""
import csv, glob, json, os, sys
root = sys.argv[1]
tot, disp = {}, {}
for path in glob.glob(os.path.join(root, "pmc*", "**", "*_counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        if "render_kernel" not in row["Kernel_Name"]:
            continue
        tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        disp.setdefault(row["Counter_Name"], set()).add((path, row["Dispatch_Id"]))
n = {k: len(v) for k, v in disp.items()}
per = {k: tot[k] / n[k] for k in tot}  # per launch
d = {}
g = per.get
if g("SQ_WAVE_CYCLES"):
    d["wave_cycles_waiting_s_waitcnt"] = g("SQ_WAIT_ANY", 0) / g("SQ_WAVE_CYCLES")
    d["wave_cycles_issue_stalled"] = g("SQ_WAIT_INST_ANY", 0) / g("SQ_WAVE_CYCLES")
    d["wave_cycles_issuing"] = g("SQ_ACTIVE_INST_ANY", 0) / g("SQ_WAVE_CYCLES")
if g("SQ_ACTIVE_INST_VALU"):
    d["valu_lane_utilisation"] = g("SQ_THREAD_CYCLES_VALU", 0) / (64.0 * g("SQ_ACTIVE_INST_VALU"))
    d["valu_active_cycles_per_instruction"] = g("SQ_ACTIVE_INST_VALU") / g("SQ_INSTS_VALU", 1)
if g("SQ_BUSY_CYCLES"):
    d["valu_active_over_busy_cycles"] = g("SQ_ACTIVE_INST_VALU", 0) / g("SQ_BUSY_CYCLES")
if g("SQ_WAVE_CYCLES") and g("GRBM_GUI_ACTIVE"):
    # SQ_WAVE_CYCLES counts in units of four clocks, summed over all waves; GRBM_GUI_ACTIVE is summed over the 8 XCDs
    d["resident_waves_per_cu"] = 4.0 * g("SQ_WAVE_CYCLES") / (g("GRBM_GUI_ACTIVE") / 8.0) / 256.0
if g("SQ_INSTS_VALU"):
    d["salu_per_valu_instruction"] = g("SQ_INSTS_SALU", 0) / g("SQ_INSTS_VALU")
if g("SQ_LDS_IDX_ACTIVE"):
    d["lds_bank_conflict_fraction"] = g("SQ_LDS_BANK_CONFLICT", 0) / g("SQ_LDS_IDX_ACTIVE")
# HBM traffic: the TCC request-size counters count 32-byte units on this part (64-byte for the
# fetch counter after its gfx950 correction of x2); both per launch.
if "WRITE_SIZE" in per:
    d["hbm_write_bytes_per_launch"] = per["WRITE_SIZE"] * 1024.0
if "FETCH_SIZE" in per:
    d["hbm_fetch_bytes_per_launch_corrected_x2"] = per["FETCH_SIZE"] * 1024.0 * 2.0
shape = None
if os.path.exists(os.path.join(root, "shape.json")):   # written by tools/profile.sh: the launch the counters belong to
    shape = json.load(open(os.path.join(root, "shape.json")))
    shape["paths_per_launch"] = shape["width"] * shape["height"] * shape["spp"]
out = {"source": root, "launch": shape, "launches_per_counter": n, "counters_per_launch": per, "derived": d}
text = json.dumps(out, indent=1)
print(text)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")
