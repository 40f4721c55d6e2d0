set -x
timeout 600 python -m pytest tests/test_gpu_engine.py -x -q -s 2>&1 | tail -30
timeout 300 python tools/host_time_engine.py 2>&1 | tail -3
GEOMAE_NO_ENGINE=1 timeout 300 python tools/host_time_engine.py 2>&1 | tail -3
timeout 300 python tools/host_time_engine.py 1 2>&1 | tail -3
GEOMAE_NO_ENGINE=1 timeout 300 python tools/host_time_engine.py 1 2>&1 | tail -3
timeout 300 python tools/host_time_engine.py 2 2>&1 | tail -3
GEOMAE_NO_ENGINE=1 timeout 300 python tools/host_time_engine.py 2 2>&1 | tail -3
