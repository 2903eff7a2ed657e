#!/bin/bash
# tools/lab/asm_loop.sh <mangled-kernel-regex>: compile conv_dma.hip to asm, print the run-length summary
# of ds_read / s_waitcnt lgkmcnt / v_mfma in the matching kernel (where the waits sit relative to the MFMAs)
F=${2:-conv_dma}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -S --cuda-device-only /root/repo/scflow_amd/csrc/$F.hip -o /tmp/$F.s || exit 1
awk "/^$1:/,/s_endpgm/" /tmp/$F.s > /tmp/k.s
grep -n "v_mfma\|s_waitcnt\|ds_read\|s_barrier\|global_load_lds" /tmp/k.s | awk -F: '{print $1": "$2}' | awk '{k=$2; if(k=="s_waitcnt")k=$2" "$3" "$4; if(k==last){c++} else {if(last!="")print first" "last" x"c; first=$1; last=k; c=1}} END{print first" "last" x"c}'
