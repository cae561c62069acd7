export PYTHONPATH=.
python tools/timing/ls_waves.py
RLS_LS_WAVES=6 python tools/timing/ls_waves.py | grep "G22\|BA-3000\|WAVES"
RLS_LS_WAVES=6 python -m pytest tests/test_gpu_local_search_fused.py -x -q 2>&1 | tail -2
python -m pytest tests/test_gpu_local_search_fused.py tests/test_gpu_shard_invariance.py -x -q 2>&1 | tail -2
