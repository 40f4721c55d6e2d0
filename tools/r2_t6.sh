GEOMAE_TEST_VERBOSE=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -s -k "forward_train_losses" 2>&1 | grep -v "^$" | tail -20
