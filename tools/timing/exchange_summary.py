"""gpurun_out/<prefix>_exchange_kt (rocprofv3 --kernel-trace of tools/timing/exchange_trace.py) + the probe's own line ->
profiles/<prefix>_exchange.json: kernel launches per exchange by name, and the measured exchange_us.

    python tools/timing/exchange_summary.py r06 [gpurun_out/r06_exchange_probe.json]
"""
import collections
import csv
import glob
import json
import os
import re
import sys

P = sys.argv[1] if len(sys.argv) > 1 else "r06"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = 100
files = glob.glob(os.path.join(ROOT, "gpurun_out", f"{P}_exchange_kt", "**", "*kernel_trace.csv"), recursive=True)
if not files:
    raise SystemExit("no kernel trace")
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name_of = lambda r: re.sub(r"\(.*", "", r["Kernel_Name"]).replace("rls::", "").replace("void ", "")
bk = [i for i, r in enumerate(rows) if "k_best_key" in r["Kernel_Name"]]
assert len(bk) in (2 * K, 3 * K), len(bk)


def window(lo, hi):
    """Every kernel dispatched from the lo-th to just before the hi-th k_best_key launch: hi - lo whole exchanges."""
    cnt, dur = collections.Counter(), collections.defaultdict(float)
    for r in rows[bk[lo]:(bk[hi] if hi < len(bk) else bk[-1] + 1)]:
        cnt[name_of(r)] += 1
        dur[name_of(r)] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    n = (hi - lo) if hi < len(bk) else (hi - lo)
    span = (int(rows[bk[min(hi, len(bk) - 1)]]["Start_Timestamp"]) - int(rows[bk[lo]]["Start_Timestamp"])) * 1e-3
    return {"exchanges": n, "kernels": {k: {"launches": c, "per_exchange": c / n, "mean_us": dur[k] / c} for k, c in cnt.items()},
            "launches_per_exchange": sum(cnt.values()) / n, "start_to_start_us": span / max(1, min(hi, len(bk) - 1) - lo)}


ex_only = window(1, K)                 # BestExchange.exchange x (K - 1)
gb = window(K + 1, 2 * K - 1)          # dist.global_best x (K - 2)
c2 = window(2 * K + 1, 3 * K - 1) if len(bk) == 3 * K else None       # dist.global_best(want_solution=True, env_offset=..) x (K - 2)
out = {
    "what": "rocprofv3 --kernel-trace of tools/timing/exchange_trace.py: 100 BestExchange.exchange calls, then 100 dist.global_best "
            "calls (exchange + rls_key_unpack), 1-rank RCCL group on one MI355X.  On a 1-rank group RCCL's in-place all_reduce "
            "dispatches no kernel of its own (at N > 1 it is one RCCL kernel): the trace shows what THIS build launches around it",
    "launches_per_exchange": ex_only["launches_per_exchange"],
    "launches_per_global_best": gb["launches_per_exchange"],
    "kernels": ex_only["kernels"],
    "BestExchange.exchange": ex_only,
    "dist.global_best": gb,
}
if c2 is not None:
    out["launches_per_global_best_with_solution"] = c2["launches_per_exchange"]
    out["dist.global_best(want_solution=True, env_offset=...)"] = c2
probe = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", f"{P}_exchange_probe.json")
if os.path.exists(probe):
    for ln in open(probe):
        if ln.startswith("EXCHANGE_PROBE "):
            out["probe"] = json.loads(ln[len("EXCHANGE_PROBE "):])
json.dump(out, open(os.path.join(ROOT, "profiles", f"{P}_exchange.json"), "w"), indent=1)
print(out["launches_per_exchange"], out["launches_per_global_best"], out.get("launches_per_global_best_with_solution"), json.dumps(out.get("probe", {})))
