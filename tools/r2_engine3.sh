set -x
GEOMAE_TEST_VERBOSE=1 timeout 900 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^\[Gloo\]" | tail -25
