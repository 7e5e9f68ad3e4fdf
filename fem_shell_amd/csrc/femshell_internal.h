/* femshell_internal.h -- shared between the translation units of libfemshell_hip.so (not installed). */
#ifndef FEMSHELL_INTERNAL_H
#define FEMSHELL_INTERNAL_H
/* records the thread-local message returned by fs_last_error(); hip_error = hipError_t or 0 */
int fs_fail_(int code, const char *what, int hip_error);
/* device copy of the quadratic-solid shape table ([ng][nn][4] then [ng] weights) on the current device */
int fs_solid_table_device_(int nn, const double **tab, int *ng);
/* compute units of the CURRENT device (cached per device; the persistent kernels launch one workgroup per CU) */
int fs_device_cu_count_(int *out);
#endif
