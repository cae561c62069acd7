#!/usr/bin/env python3
"""SQ counter evidence for the kernels whose bound is on chip (rocprofv3 --pmc passes of tools/sweeps/pmc_passes.sh over
tools/bench_configs.py --profile):

    python tools/sq_table.py r03 gpurun_out/r03_sq   ->  profiles/r03_sq_counters.md / .json

Per kernel group (name, grid, workgroup): means per launch of the counters collected, and the ratios that say what a
wave's (quad-)cycles were spent on: VALU = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (the share of its resident cycles in
which a wave has a VALU instruction executing: times the resident waves per SIMD = that SIMD's VALU utilisation),
wait-inst = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (waiting for an instruction's operands / a counter), wait-any =
SQ_WAIT_ANY / SQ_WAVE_CYCLES (incl. barriers and instruction-buffer stalls), LDS = SQ_ACTIVE_INST_LDS / SQ_WAVE_CYCLES,
LDS bank-conflict cycles / LDS active cycles, VALU instructions per wave."""
import collections
import csv
import glob
import json
import os
import re
import sys

tag, src = sys.argv[1], sys.argv[2]
WANT = [r"k_maxcut_greedy_sweep_levels", r"k_maxcut_propose_accept32", r"k_mcpg_local_search_levels", r"k_maxcut_local_search", r"k_mcpg_metro_packed", r"k_isco_maxcut_step",
        r"k_isco_tsp_step", r"k_qubo_ls_value_mfma", r"k_qubo_sparse_ls_value", r"k_maxcut_obj<", r"k_maxcut_obj32<", r"k_node_stats_bits", r"k_maxcut_step<",
        r"k_spin_step", r"k_tsp_2opt", r"k_ls_threshold<", r"k_ls_mask<", r"k_ls_propose<", r"k_ls_apply_rounds<", r"k_ls_apply_rounds32<"]


def short(n):
    n = n.replace("void rls::", "").replace("rls::", "")
    return re.sub(r"\(.*$", "", n)[:100]


acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"{src}/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        name = short(r["Kernel_Name"])
        if not any(re.search(w, name) for w in WANT):
            continue
        k = (name, int(r["Grid_Size"]), int(r["Workgroup_Size"]))
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for (name, grid, wg), c in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    m = {k: sum(v) / len(v) for k, v in c.items()}
    rec = {"kernel": name, "grid": grid, "workgroup": wg, "launches": max(len(v) for v in c.values()), "counters": m}
    g = m.get
    if g("SQ_WAVE_CYCLES"):
        if g("SQ_WAIT_INST_ANY") is not None:
            rec["wait_share_of_wave_cycles"] = g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")
        if g("SQ_ACTIVE_INST_LDS") is not None:
            rec["lds_share_of_wave_cycles"] = g("SQ_ACTIVE_INST_LDS") / g("SQ_WAVE_CYCLES")
        if g("SQ_ACTIVE_INST_VALU") is not None:
            rec["valu_share_of_wave_cycles"] = g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES")
        if g("SQ_WAIT_ANY") is not None:
            rec["wait_any_share_of_wave_cycles"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
    if g("SQ_BUSY_CYCLES") and g("SQ_ACTIVE_INST_VALU") is not None:
        rec["valu_active_per_busy_cycle"] = g("SQ_ACTIVE_INST_VALU") / g("SQ_BUSY_CYCLES")
    if g("SQ_ACTIVE_INST_LDS") and g("SQ_LDS_BANK_CONFLICT") is not None:
        rec["lds_bank_conflict_share"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_ACTIVE_INST_LDS")
    if g("SQ_WAVES") and g("SQ_INSTS_VALU") is not None:
        rec["valu_insts_per_wave"] = g("SQ_INSTS_VALU") / g("SQ_WAVES")
    rows.append(rec)
os.makedirs("profiles", exist_ok=True)
json.dump({"note": __doc__.strip(), "source": src, "groups": rows}, open(f"profiles/{tag}_sq_counters.json", "w"), indent=1)
with open(f"profiles/{tag}_sq_counters.md", "w") as f:
    f.write(f"# {tag}: SQ counters of the on-chip-bound kernels (tools/sweeps/pmc_passes.sh -> tools/sq_table.py)\n\n")
    f.write("Means per launch; separate `rocprofv3 --pmc` passes, never combined with trace domains.  Shares are of SQ_WAVE_CYCLES "
            "(quad-cycles a wave is resident): VALU = SQ_ACTIVE_INST_VALU, wait-inst = SQ_WAIT_INST_ANY, wait-any = SQ_WAIT_ANY, "
            "LDS = SQ_ACTIVE_INST_LDS; conflicts = SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS.  VALU share x resident waves per SIMD "
            "= the SIMD's VALU utilisation.\n\n")
    f.write("| kernel | grid x wg | waves | wave cycles | VALU | wait-inst | wait-any | LDS | LDS conflicts | VALU insts / wave | SALU insts | LDS insts |\n|---|---|---|---|---|---|---|---|---|---|---|---|\n")
    for r in rows:
        m = r["counters"]
        fmt = lambda k, p="{:.2f}": p.format(r[k]) if k in r else ""
        f.write("| `{}` | {} x {} | {} | {} | {} | {} | {} | {} | {} | {} | {} | {} |\n".format(
            r["kernel"][:80], r["grid"], r["workgroup"], f"{m.get('SQ_WAVES', 0):.0f}", f"{m.get('SQ_WAVE_CYCLES', 0):.3g}",
            fmt("valu_share_of_wave_cycles"), fmt("wait_share_of_wave_cycles"), fmt("wait_any_share_of_wave_cycles"), fmt("lds_share_of_wave_cycles"),
            fmt("lds_bank_conflict_share"), fmt("valu_insts_per_wave", "{:.0f}"), f"{m.get('SQ_INSTS_SALU', 0):.3g}", f"{m.get('SQ_INSTS_LDS', 0):.3g}"))
print(open(f"profiles/{tag}_sq_counters.md").read())
