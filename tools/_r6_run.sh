python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^$" | grep -E "passed|failed|FAILED|Error|error|assert|loss err|plain-bf16|grad" | head -80
