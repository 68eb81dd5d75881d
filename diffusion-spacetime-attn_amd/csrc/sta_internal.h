// Internal to libsta_xattn.so: the thread-local error text behind sta_last_error().
#ifndef STA_INTERNAL_H
#define STA_INTERNAL_H
extern thread_local char g_sta_err[256];
int sta_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
#endif
