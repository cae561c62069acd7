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
cnt, dur = collections.Counter(), collections.defaultdict(float)
for f in files:
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("rls::", "")
        cnt[name] += 1
        dur[name] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
# the stretch of K BestExchange.exchange calls + K global_best calls: every kernel whose count is a multiple of K belongs to it
kern = {}
for name, c in sorted(cnt.items(), key=lambda kv: -kv[1]):
    kern[name] = {"launches": c, "mean_us": dur[name] / c}
per_exchange = {n: v["launches"] for n, v in kern.items()}
best_key = sum(c for n, c in per_exchange.items() if "k_best_key" in n)
unpack = sum(c for n, c in per_exchange.items() if "k_key_unpack" in n)
coll = sum(c for n, c in per_exchange.items() if "ccl" in n.lower() or "allreduce" in n.lower() or "AllReduce" in n)
other = {n: c for n, c in per_exchange.items() if not ("k_best_key" in n or "k_key_unpack" in n or "ccl" in n.lower() or "allreduce" in n.lower())}
out = {
    "what": "rocprofv3 --kernel-trace of tools/timing/exchange_trace.py: 100 BestExchange.exchange calls, then 100 dist.global_best "
            "calls (exchange + rls_key_unpack), 1-rank RCCL group on one MI355X",
    "launches_per_exchange": {"rls_best_key": best_key / (2 * K), "collective_kernels": coll / (2 * K),
                              "rls_key_unpack (global_best only, after the collective)": unpack / K},
    "kernels": kern,
    "other_kernels_in_the_whole_trace (set-up: arange / mul / remainder / to of the test vector, barrier)": other,
}
probe = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", f"{P}_exchange_probe.json")
if os.path.exists(probe):
    for ln in open(probe):
        if ln.startswith("EXCHANGE_PROBE "):
            out["probe"] = json.loads(ln[len("EXCHANGE_PROBE "):])
json.dump(out, open(os.path.join(ROOT, "profiles", f"{P}_exchange.json"), "w"), indent=1)
print(json.dumps(out["launches_per_exchange"]), json.dumps(out.get("probe", {})))
