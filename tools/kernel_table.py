#!/usr/bin/env python3
"""Per-kernel roofline table of a round from the three rocprofv3 passes of tools/profile_round.sh over
tools/bench_configs.py (kernel trace; --pmc FETCH_SIZE; --pmc WRITE_SIZE):

    python tools/kernel_table.py r02 [--src gpurun_out/r02a]   ->  profiles/r02_kernels.json, profiles/r02_kernels.md

Dispatches are grouped by (kernel, grid size, workgroup size, LDS): the same kernel runs at several BASELINE configs
in one bench_configs pass.  Per group: launches, mean duration (kernel trace), HBM bytes per launch from the PMC
passes with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE tallies 64 B per 128-B request of a wide
coalesced read -> read bytes = 2 x FETCH_SIZE; WRITE_SIZE exact for 16-B-per-lane stores; both in KB of 1024 B),
and -- where the group is one of the rows below -- the ALGORITHMIC bytes per launch (SURVEY.md section 8d x the units
one launch processes) and achieved / 8 TB/s.

Kernels whose bound is on chip get a second roofline, the VALU ISSUE roofline, from the SQ passes of the same command
(profiles/<tag>_sq_counters.json, tools/sq_table.py): SQ_INSTS_VALU wave-instructions per launch spread over the chip's 1024
SIMDs.  One wave's own stream issues a vector instruction every 4 cycles (MI355X_MICROARCH.md, "vector-instruction ISSUE
cost"); with two or more waves resident a SIMD-32 retires a wave64 instruction every 2.  `valu floor us` is the 4-cycle
figure at 2.4 GHz -- instructions x 4 / (1024 x 2.4e9) -- and `valu frac` = floor / mean duration: 1.0 means every SIMD
issued from one stream without a gap for the whole launch; values above 1 are possible (and mean the waves of a SIMD
overlapped their issue) up to 2.0, the 2-cycle bound.  Measured on this pool (tools/ceilings/valu_issue.hip, independent
register-only instructions, cycles per wave-instruction per SIMD at the nominal 2.4 GHz): one wave per SIMD 4.8-5.5 whatever
the instruction; with two or four waves 2.4-2.6 for v_xor / v_and / v_bitop3 / v_add_u32 / v_mov / v_lshrrev / v_ashrrev /
v_fma_f32 / v_mul_f32 (valu frac 1.5-1.65), 4.2-4.3 for v_lshlrev / v_lshl_or / v_and_or / v_or3 / v_bfi / v_bfe / v_perm /
v_alignbit / v_bcnt / v_mul_lo / v_mul_u32_u24 / v_mad_u32_u24 / v_add3 / v_add_co / v_cvt and the 64-bit shift (0.93), 8.2
for v_log / v_sin / v_sqrt (0.49) -- so the ceiling of a bit-sliced counter kernel (mostly bitop3 / xor / and) is ~1.5,
that of a hash- or transpose-heavy one (multiplies, perm, alignbit) nearer 1.  A kernel far below 1 on BOTH rooflines is
waiting (barriers, latency): its lever is neither bytes nor instruction count.
"""
import argparse
import collections
import csv
import json
import os
import re

HBM = 8e12

# (kernel regex, grid threads, units B per launch, algorithmic bytes per unit, label).  The grid pins the config: tile
# kernels launch B / 64 workgroups, the step kernel B / EPW waves, wave-per-env kernels B waves.
N22, N70, N14, NT, NBA = 2000, 10000, 800, 100, 10000
T22, T70 = 65536 // 64, 131072 // 64          # 64-env tiles
ROWS = [
    (r"k_maxcut_step<unsigned char, 4, 2, true", 65536 // 4 * 64, 65536, 2 * N22 + 20, "K4 maxcut_step emit u8 | G22 2^16 (headline)"),
    (r"k_maxcut_step<float, 1, [23], true", 65536 * 64, 65536, 8 * N22 + 20, "K4 maxcut_step emit f32 gym surface | G22 2^16 (round 4: stores chase the loads)"),
    (r"k_maxcut_step<unsigned char, 1, 2, true", 131072 * 64, 131072, 2 * N70 + 20, "K4 maxcut_step emit u8 | G70 2^17"),
    (r"k_maxcut_step<float, 1, [23], true", 131072 * 64, 131072, 8 * N70 + 20, "K4 maxcut_step emit f32 gym surface | G70 2^17 (80 KB of LDS per workgroup, nontemporal stores)"),
    (r"k_maxcut_step<unsigned char, 8, 2, true", 256 // 8 * 64, 256, 2 * N14 + 20, "K4 maxcut_step emit u8 | G14 256 (launch-bound)"),
    (r"k_maxcut_obj<", T22 * 256, 65536, N22 + 8, "K1 maxcut_obj | G22 2^16"),
    (r"k_maxcut_obj<", T70 * 512, 131072, N70 + 8, "K1 maxcut_obj | G70 2^17"),
    (r"k_maxcut_obj32<", 2 * T70 * 256, 131072, N70 + 8, "K1 maxcut_obj on half tiles (32 envs per workgroup: rows past 8192 nodes) | G70 2^17"),
    (r"k_maxcut_propose_accept<true, \d+, \d+, true>", T22 * 256, 65536, 2 * N22 + N22 // 8 + 16, "K6 propose_accept, bit-packed mask | G22 2^16"),
    (r"k_maxcut_propose_accept<true, \d+, \d+, true>", T70 * 512, 131072, 2 * N70 + N70 // 8 + 16, "K6 propose_accept, bit-packed mask | G70 2^17"),
    (r"k_maxcut_propose_accept<", T22 * 256, 65536, 2 * N22 + 16, "K6 propose_accept, byte mask | G22 2^16"),
    (r"k_maxcut_propose_accept<", T70 * 512, 131072, 2 * N70 + 16, "K6 propose_accept, byte mask | G70 2^17"),
    (r"k_maxcut_greedy_sweep_levels<", T22 * 256, 65536, 2 * N22 + 16, "K5 greedy_sweep | G22 2^16 (on-chip bound)"),
    (r"k_maxcut_greedy_sweep_levels<", T70 * 512, 131072, 2 * N70 + 16, "K5 greedy_sweep | G70 2^17 (on-chip bound)"),
    (r"k_maxcut_greedy_sweep_levels32<", 2 * T70 * 512, 131072, 2 * N70 + 16, "K5 greedy_sweep on half tiles (rows past 8192 nodes) | G70 2^17 (on-chip bound)"),
    (r"k_maxcut_propose_accept32<true, \d+, \d+, true>", 2 * T70 * 512, 131072, 2 * N70 + N70 // 8 + 16, "K6 propose_accept on half tiles, bit-packed mask | G70 2^17"),
    (r"k_maxcut_propose_accept32<", 2 * T70 * 512, 131072, 2 * N70 + 16, "K6 propose_accept on half tiles, byte mask | G70 2^17"),
    (r"k_node_stats_bits<1", T22 * 512, 65536, 5 * N22, "K3 delta_all | G22 2^16"),
    (r"k_node_stats_bits32<1", 2 * T22 * 512, 65536, 5 * N22, "K3 delta_all on half tiles (short rows at a full batch) | G22 2^16"),
    (r"k_node_stats_bits<1", T70 * 512, 131072, 5 * N70, "K3 delta_all | G70 2^17"),
    (r"k_node_stats_bits<2, true, false, signed char", T22 * 512, 65536, 2 * N22, "ls_weights pre-pass, int8 weights + batch min / max | G22 2^16"),
    (r"k_node_stats_bits<2, true, false, signed char", 64 * 512, 4096, 2 * N22, "ls_weights pre-pass, int8 weights + batch min / max | G22 4096"),
    (r"k_maxcut_local_search<true, signed char, 16, 4>", T22 * 256, 65536, 3 * N22 + 8, "LS fused local search: threshold + 8 proposal rounds + sweep | G22 2^16 (VALU-bound; x in, int8 ws, x out)"),
    (r"k_maxcut_local_search<true, signed char, 16, 8>", 64 * 512, 4096, 3 * N22 + 8, "LS fused local search: threshold + 8 proposal rounds + sweep | G22 4096 (64 workgroups: latency / VALU)"),
    (r"k_ls_propose<short, 24, true, false>", 512 * 512, 32768, 4 * NBA + 16, "LS round kernel: noise + mask + count + accept (one proposal round per launch) | BA-1e4 2^15 (VALU-bound; x in, int16 ws in, accepted rows out)"),
    (r"k_ls_threshold<short>", 512 * 512, 32768, 2 * NBA + 4, "LS threshold kernel: noise + top-9 per env | BA-1e4 2^15 (VALU-bound; int16 ws in)"),
    (r"k_ls_mask<short>", 4 * 64 * 512, 4096, 2 * NBA + NBA // 8, "LS mask kernel: noise + mask for a quarter of the rows per workgroup | BA-1e4 4096 (64 tiles x 4 slices)"),
    (r"k_ls_apply_rounds<24, 8>", 64 * 512, 4096, 2 * NBA + 8 * (NBA // 8) * 8 + 16, "LS apply kernel: 8 rounds of (x ^ mask words, count, accept, undo) on one load of the tile | BA-1e4 4096"),
    (r"k_ls_apply_rounds32<24, 8>", 128 * 512, 4096, 2 * NBA + 8 * (NBA // 8) * 8 + 16, "LS apply kernel on half tiles (a batch of few tiles: 128 workgroups instead of 64): 8 rounds on one load of the tile | BA-1e4 4096"),
    (r"k_node_stats_bits<2, true, true, short", 512 * 512, 32768, 3 * NBA, "ls_weights pre-pass, int16 weights (hub graph: 16 counter planes) | BA-1e4 2^15"),
    (r"k_tsp_tour_length", None, 65536, 8 * NT + 4, "K12 tsp_tour_length | TSP-100 2^16"),
    (r"k_tsp_swap_delta_all<true, true, ", None, 65536, 21 * NT, "K13 tsp_swap_delta_all = ISCO_TSP.opt_2, partners drawn in the kernel | TSP-100 2^16 (LDS-gather-bound)"),
    (r"k_tsp_swap_delta_all<true, false, ", None, 65536, 29 * NT, "K13 tsp_swap_delta_all, selected [B, N] given (test hook) | TSP-100 2^16"),
    (r"k_rand_spins_multi<16>", 1048576, 65536, N22, "K14 rand_spins (reset path) | G22 2^16 (write-only)"),
    (r"k_rand_spins_multi<16>", 2097152, 131072, N70, "K14 rand_spins (reset path) | G70 2^17 (write-only)"),
    (r"k_rand_perms_lds", None, 65536, 8 * NT, "K14 rand_perms (reset path) | TSP-100 2^16 (write-only; Fisher-Yates chain per lane)"),
    (r"k_qubo_sparse_levels<", None, None, None, None),
    (r"k_best_key<int>", None, 65536, 4, "C1 rls_best_key: argmax + packed key of one rank's objectives | 2^16 envs (one workgroup: latency)"),
    (r"k_spin_step<float, true, false>", 16384 * 64, 16384, None, "S1 spin_step | G22-sized 2^14: O(deg) per env, nothing streamed (round 2: 24 N bytes per env-step, 250 us)"),
    (r"k_spin_step<float, true, false>", 4096 * 64, 4096, None, "S1 spin_step | BA-200 4096: O(deg)"),
    (r"k_spin_step<float, true, true>", 1024 * 64, 1024, 16 * 200, "S1d spin_step_dense | per-env BA-200 matrices, 1024 envs (launch-bound; the flipped node's matrix row + the entries it changes)"),
    (r"k_spin_observation_rows<float, true>", 2 * 256 * 16384, 16384, 40 * N22, "S1 observation, rows only [B, 7, N] | G22-sized 2^14 (7 rows written, spins + immediate + last-flip read)"),
    (r"k_spin_observation_rows<float, true>", 1 * 256 * 4096, 4096, 40 * 200, "S1 observation, rows only [B, 7, N] | BA-200 4096"),
    (r"k_mcpg_pack_f32x4", 41156608, 262144, 4 * NBA + NBA // 8, "f32 [N, C] -> bit-packed chains (the shim of the reference-shaped MCPG calls) | BA-1e4 2^18"),
    (r"k_mcpg_unpack<true, true, 2>", 327680000, 262144, 4 * NBA + NBA // 8, "bit-packed chains -> f32 [N, C] (the shim of the reference-shaped MCPG calls) | BA-1e4 2^18"),
    (r"k_spin_observation<float, true>", 41 * 1024 * 256, 1024, 4 * (207 * 200 + 7 * 200 + 200 * 200), "S1d observation [B, 7+N, N] with per-env matrix rows | BA-200 1024"),
    (r"k_rand_couplings_ba<float>", 128 * 64, 1024, 4 * 200 * 200, "rand_couplings BA (m=4) | 1024 x BA-200 (a dependent chain per env: latency-bound)"),
    (r"k_qubo_ls_value", None, None, None, None),
    (r"k_mcpg_local_search_levels<float, Packed64", None, 262144, 4 * NBA + NBA // 8, "K7+K8 local_search_levels, f32 [N,C] in -> packed out | BA-1e4 2^18"),
    (r"k_mcpg_local_search_levels<Packed64, Packed64", None, 262144, 2 * (NBA // 8), "K7+K8 local_search_levels, bit-packed in place | BA-1e4 2^18 (VALU-bound)"),
    (r"k_mcpg_metro_packed", None, 262144, 2 * (NBA // 8), "K9 metro rounds, bit-packed (1000 rounds per launch) | BA-1e4 2^18 (latency-bound walk)"),
    (r"k_mcpg_pick_gather_packed", None, 2048, 64 * (NBA // 8), "K8b best-of-repeats gather, bit-packed | 2048 kept chains (reads 64 tiles per kept tile)"),
    (r"k_mcpg_value_bit_sums", None, 262144, NBA // 8, "get_return bit sums | BA-1e4 2^18"),
    (r"k_isco_maxcut_step", None, 4096, 8 * N22, "I1 ISCO_maxcut.step | G22-sized, 4096 samples (f32 sample in, f32 sample out)"),
    (r"k_isco_tsp_step", None, 65536, 16 * NT, "I2 ISCO_TSP.step, 8 rounds | TSP-100 2^16 (int64 tour in, int64 tour out)"),
]
CLOCK_HZ, SIMDS = 2.4e9, 1024


def short(n):
    n = n.replace("void rls::", "").replace("rls::", "")
    return re.sub(r"\(.*$", "", n)[:120]


def load_trace(path):
    g = collections.defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(path)):
        k = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]),
             int(r["Workgroup_Size_X"]), int(r["LDS_Block_Size"]))
        g[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        meta[k] = {"vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"]), "scratch": int(r["Scratch_Size"])}
    return g, meta


def load_pmc(path, name):
    g = collections.defaultdict(list)
    if not os.path.exists(path):
        return g
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        k = (short(r["Kernel_Name"]), int(r["Grid_Size"]), int(r["Workgroup_Size"]), int(r["LDS_Block_Size"]))
        g[k].append(float(r["Counter_Value"]))
    return g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--src", default=None, help="prefix of the gpurun_out directories (default gpurun_out/<tag>)")
    ap.add_argument("--what", default="cfg")
    ap.add_argument("--sq", default=None, help="profiles/<tag>_sq_counters.json of the same command (default: that path if it exists)")
    a = ap.parse_args()
    sq_path = a.sq or f"profiles/{a.tag}_sq_counters.json"
    sq = {}
    if os.path.exists(sq_path):
        for g in json.load(open(sq_path))["groups"]:
            v = g["counters"].get("SQ_INSTS_VALU")
            if v:
                sq[(g["kernel"][:100], g["grid"], g["workgroup"])] = g
    src = a.src or f"gpurun_out/{a.tag}"
    pfx = os.path.basename(src)
    trace, meta = load_trace(f"{src}_{a.what}_kt/{pfx}_kernel_trace.csv")
    fetch = load_pmc(f"{src}_{a.what}_fetch/{pfx}_counter_collection.csv", "FETCH_SIZE")
    write = load_pmc(f"{src}_{a.what}_write/{pfx}_counter_collection.csv", "WRITE_SIZE")
    # K6 writes its accepted rows: the accept rate of the timed loop is measured by tools/bench_configs.py (in the row's note) and those
    # N bytes per accepted proposal are algorithmic bytes of the launch
    k6_rate = {}
    cfg_path = f"profiles/{a.tag}_configs_profiled.jsonl"
    if os.path.exists(cfg_path):
        for line in open(cfg_path):
            try:
                r = json.loads(line)
            except ValueError:
                continue
            m = re.search(r"accept rate of this loop ([0-9.]+)", r.get("note", ""))
            if m and r.get("kernel", "").startswith("K6"):
                k6_rate[(r["config"][:3], "bit-packed" in r["kernel"])] = float(m.group(1))
    out = []
    for k, durs in sorted(trace.items(), key=lambda kv: -sum(kv[1])):
        name, grid, wg, lds = k
        if name.startswith("__amd") or "at::native" in name or "elementwise" in name or "hipcub" in name or "rocprim" in name \
                or name.strip() in ("void", ""):
            continue
        d = sorted(durs)
        # steady state: drop the first (cold) launch of a group when there are several
        use = durs[1:] if len(durs) > 2 else durs
        rec = {"kernel": name, "grid": grid, "workgroup": wg, "lds_bytes": lds, "launches": len(durs),
               "mean_us": sum(use) / len(use) / 1e3, "min_us": d[0] / 1e3, **meta[k]}
        f, w = fetch.get(k), write.get(k)
        if f:
            rec["read_bytes"] = 2 * 1024 * sum(f) / len(f)
        if w:
            rec["write_bytes"] = 1024 * sum(w) / len(w)
        if f and w:
            rec["hbm_bytes"] = rec["read_bytes"] + rec["write_bytes"]
        for pat, want_grid, B, per_unit, label in ROWS:
            if label is None or re.search(pat, name) is None or (want_grid is not None and want_grid != grid):
                continue
            rec.update({"row": label, "units_per_launch": B})
            if per_unit is not None and label.startswith("K6"):
                size = "G22" if "G22" in label else "G70"
                rate = k6_rate.get((size, "bit-packed" in label))
                if rate is not None:
                    per_unit = per_unit + rate * (N22 if size == "G22" else N70)
                    rec["row"] = label + f" (accept rate {rate:.2f}: those rows are written)"
            if per_unit is not None:
                rec.update({"algorithmic_bytes": B * per_unit, "achieved_GBps": B * per_unit / (rec["mean_us"] * 1e-6) / 1e9,
                            "frac_of_8TBps": B * per_unit / (rec["mean_us"] * 1e-6) / HBM})
                if "hbm_bytes" in rec:
                    rec["traffic_over_algorithmic"] = rec["hbm_bytes"] / (B * per_unit)
            break
        g = sq.get((name[:100], grid, wg))
        if g is not None:
            insts = g["counters"]["SQ_INSTS_VALU"]
            rec["valu_insts"] = insts
            rec["valu_floor_us"] = insts * 4 / (SIMDS * CLOCK_HZ) * 1e6
            rec["valu_frac"] = rec["valu_floor_us"] / rec["mean_us"]
            for k in ("valu_share_of_wave_cycles", "wait_any_share_of_wave_cycles", "lds_bank_conflict_share"):
                if k in g:
                    rec[k] = g[k]
        out.append(rec)
    os.makedirs("profiles", exist_ok=True)
    json.dump({"note": __doc__.strip().split("\n\n")[1], "source": f"{src}_{a.what}_*", "groups": out},
              open(f"profiles/{a.tag}_kernels.json", "w"), indent=1)
    with open(f"profiles/{a.tag}_kernels.md", "w") as f:
        f.write(f"# {a.tag}: per-kernel rocprofv3 summary (tools/profile_round.sh -> tools/kernel_table.py)\n\n")
        f.write("mean us = kernel-trace duration (first launch of a group dropped); read = 2 x FETCH_SIZE, write = WRITE_SIZE "
                "(separate --pmc passes, KB of 1024 B); frac = algorithmic bytes / mean / 8 TB/s\n\n")
        f.write("valu floor us = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x 2.4 GHz), valu frac = floor / mean us (the VALU-issue roofline of the "
                "on-chip-bound kernels: 1.0 = every SIMD issuing from one stream for the whole launch, 2.0 = the SIMD-32 bound with >= 2 waves; measured ceiling with >= 2 waves per SIMD: ~1.5 for "
                "bitop3 / xor / and / add streams, 0.93 for perm / alignbit / bfe / lshl_or / integer-multiply streams, tools/ceilings/valu_issue.hip); "
                "wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES; conflicts = SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS\n\n")
        f.write("| kernel | grid x wg | LDS | VGPR | launches | mean us | read MB | write MB | row | alg MB | frac | traffic/alg | valu floor us | valu frac | wait | conflicts |\n")
        f.write("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for r in out:
            f.write("| `{}` | {} x {} | {} | {} | {} | {:.1f} | {} | {} | {} | {} | {} | {} | {} | {} | {} | {} |\n".format(
                r["kernel"][:90], r["grid"], r["workgroup"], r["lds_bytes"], r["vgpr"], r["launches"], r["mean_us"],
                f"{r['read_bytes'] / 1e6:.1f}" if "read_bytes" in r else "", f"{r['write_bytes'] / 1e6:.1f}" if "write_bytes" in r else "",
                r.get("row", ""), f"{r['algorithmic_bytes'] / 1e6:.1f}" if "algorithmic_bytes" in r else "",
                f"{r['frac_of_8TBps']:.3f}" if "frac_of_8TBps" in r else "",
                f"{r['traffic_over_algorithmic']:.2f}" if "traffic_over_algorithmic" in r else "",
                f"{r['valu_floor_us']:.1f}" if "valu_floor_us" in r else "", f"{r['valu_frac']:.2f}" if "valu_frac" in r else "",
                f"{r['wait_any_share_of_wave_cycles']:.2f}" if "wait_any_share_of_wave_cycles" in r else "",
                f"{r['lds_bank_conflict_share']:.2f}" if "lds_bank_conflict_share" in r else ""))
    print(open(f"profiles/{a.tag}_kernels.md").read())


if __name__ == "__main__":
    main()
