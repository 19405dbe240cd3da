// fir_pf_table.h -- which fir_oa_kernel variant each block height (8 ... 15 rows
// of 256 samples) runs by default: 0 = spectrum resident, 1 = next pair's
// samples requested ahead (FirPair::nx), 2 = samples and spectrum ahead
// (FirPair::Hn).  Variants 1 and 2 keep loads in flight in registers the
// compiler knows nothing about; an entry may only name a variant whose
// assembly benchmarks/check_async_regions.py finds clean (tests/test_fir_async.py
// rebuilds the assembly and checks exactly that).  Measured on the BASELINE chunk
// (1024 taps, 12 rows), same box: 0: 1.078 ms, 1: 1.048-1.052 ms, 2: 1.037-1.044 ms -- variant 2
// moves the stalls (fir_stamps) and gains under 1 %, so 1 is the default everywhere;
// OSZ_FIR_PF=0|1|2 overrides the table.
#pragma once
#define OSZ_FIR_PF_TABLE {1, 1, 1, 1, 1, 1, 1, 1}
