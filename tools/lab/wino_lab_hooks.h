// Compile-time phase ablations of conv_wino_kernel for tools/lab/wino_phases.py (never part of the library
// build; scflow_amd/csrc/conv_wino.hip includes this file only under -DSCF_WINO_LAB, see
// tools/lab/build_wino_masks.sh).  SCF_WINO_LAB_MASK bits: 0 no MFMAs, 1 no input transform, 2 no copies in
// the chunk loop, 3 no output stores (and nothing after the pair exchange), 4 no per-chunk barrier, 5 no U reads
// in the chunk loop.  -DSCF_WINO_DEFAULT_VARIANT=2|3 makes the quarter-domain kernel (4 / 8 waves) the default.
// Results are wrong with any bit set; only the durations mean something.
#pragma once
#ifndef SCF_WINO_LAB_MASK
#define SCF_WINO_LAB_MASK 0
#endif
#define WN_LAB(bit) ((SCF_WINO_LAB_MASK >> (bit)) & 1)
#define WN_LAB_FIELDS
