/*
 * rt_tables.h -- constant numeric tables (blue-noise tile, disk samples, constant spectra,
 * spectrum->RGB matrix) embedded from ray-tracer_amd/data/tables.bin at build time.
 * The numbers come from the reference's data files (BlueNoiseTile.cpp:4-1605, LowDiscrepancy.cpp:4-258,
 * RadiometryColor.cpp:26-37, :41-945) through tools/extract_reference_tables.py.
 */
#pragma once
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
enum { RT_TABLE_F32 = 0, RT_TABLE_F64 = 1, RT_TABLE_I32 = 2 };
/* returns a pointer to the table payload or 0; dtype/count may be null */
const void *rt_table_find(const char *name, int *dtype, int *count);
/* i-th table name, 0 past the end */
const char *rt_table_name(int i);
#ifdef __cplusplus
}
#endif
