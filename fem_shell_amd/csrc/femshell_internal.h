/* femshell_internal.h -- shared between the translation units of libfemshell_hip.so (not installed). */
#ifndef FEMSHELL_INTERNAL_H
#define FEMSHELL_INTERNAL_H
/* records the thread-local message returned by fs_last_error(); hip_error = hipError_t or 0 */
int fs_fail_(int code, const char *what, int hip_error);
#endif
