python -m pytest tests/test_gpu_graph.py -q -x -p no:cacheprovider 2>&1 | tail -3
