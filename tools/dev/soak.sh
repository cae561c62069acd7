export PYTHONPATH=.
for f in maxcut ls gym spin mcpg mcpg_round isco isco_tsp tsp qubo select rand; do
  timeout 260 python tools/fuzz/fuzz_$f.py 200 20261005 2>&1 | tail -1
done
