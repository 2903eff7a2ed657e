#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 300 tools/lab/bin/lookup_lab 32 20 32 32 > $O/lab_b32.txt 2>&1; grep -v "wave->\|levels per\|TG_ID" $O/lab_b32.txt | cut -c1-250 | head -24
timeout 300 tools/lab/bin/lookup_lab 8 20 60 80 > $O/lab_c4.txt 2>&1; grep -v "wave->\|levels per\|TG_ID\|  level" $O/lab_c4.txt | cut -c1-250 | head -12
timeout 300 python tools/lab/order_ab.py 2>&1 | tail -5
