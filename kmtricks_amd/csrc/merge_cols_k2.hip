// merge_cols_k2.hip -- the column-blocked merge (merge_cols.hip) compiled for 128-bit keys (32 <= k <= 63, Kmer<64>): 5-dword records,
// 8 window slots per lane, one row table of 32-byte entries.  Same kernels, same results; entry points in namespace kmx::cols_k2.
#define KMX_CL_KW 2
#include "merge_cols.hip"
