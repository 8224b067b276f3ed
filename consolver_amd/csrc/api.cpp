// error plumbing + version queries of the C ABI
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void cs_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {
int cs_abi_version(void) { return CS_ABI_VERSION; }
const char* cs_last_error(void) { return g_err; }
const char* cs_target_arch(void) { return "gfx950"; }
const char* cs_error_string(int code) {
    switch (code) {
        case CS_OK: return "ok";
        case CS_E_ARG: return "invalid argument";
        case CS_E_SHAPE: return "invalid shape";
        case CS_E_DTYPE: return "unsupported dtype";
        case CS_E_HIP: return "HIP runtime error";
        case CS_E_STATE: return "invalid state";
        case CS_E_UNSUPPORTED: return "not implemented";
        default: return "unknown error";
    }
}
}
