#!/bin/bash
timeout 900 python -m pytest tests/test_golden_gpu.py -x -q -s -k "G6" 2>&1 | grep -E "G6:|passed|failed|Error|assert|Mismatch|Max abs|Max rel" | head -20
