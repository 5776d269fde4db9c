// merge_rows_small.hip -- k_merge_rows for cohorts of up to 256 lists: merge_rows.hip built with 512 threads and 2048 record slots a
// workgroup, two workgroups a CU (round 6; see the note at the top of merge_rows.hip)
#define KMX_ROWS_SMALL 1
#define KMX_ROWS_TPB 512
#define KMX_ROWS_CAP 2048
#include "merge_rows.hip"
