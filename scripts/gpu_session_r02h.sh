#!/bin/bash
mkdir -p gpurun_out/r02h
timeout 600 python scripts/debug_train_grads.py 2>&1 | grep -v Warning | tail -20
