// Last-error string of the C-ABI (thread-local).
#include <stdarg.h>
#include <stdio.h>
static thread_local char g_err[1024] = "";
void mc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* mc_last_error(void) { return g_err; }
