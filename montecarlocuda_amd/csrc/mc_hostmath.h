/* mc_hostmath.h -- internal: what mc_api.hip / host_path.c take from mc_hostmath.c besides the public symbols. */
#ifndef MC_HOSTMATH_H_
#define MC_HOSTMATH_H_
#ifdef __cplusplus
extern "C" {
#endif
/* Records the calling thread's error text (mc_last_error) and returns `code`. */
int mc_internal_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
#ifdef __cplusplus
}
#endif
#endif
