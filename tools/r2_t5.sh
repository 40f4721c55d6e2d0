set -x
GEOMAE_TEST_VERBOSE=1 timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -s 2>&1 | grep -v "^$" | tail -40
