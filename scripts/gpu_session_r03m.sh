#!/bin/bash
timeout 300 python scripts/rows_mlp_bench.py 2>&1 | grep -v amdgpu
