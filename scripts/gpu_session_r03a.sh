#!/bin/bash
timeout 600 python -m pytest tests/test_syncbn_gpu.py tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -15
