python -m pytest tests/test_gpu_wide.py -x -q 2>&1 | tail -15
python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_wide.py 2>&1 | tail -3
