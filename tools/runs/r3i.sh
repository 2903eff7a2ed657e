#!/bin/bash
# Winograd kernel: first correctness run + timing of the eligible layers (direct vs Winograd)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -q -k "winograd" -s 2>&1 | tail -25 > gpurun_out/r3i_test.txt
timeout 600 python tools/lab/wino_layers.py > gpurun_out/r3i_layers.txt 2>&1
cat gpurun_out/r3i_test.txt gpurun_out/r3i_layers.txt
