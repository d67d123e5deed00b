// Shared host-side helpers for the gfx950 kernels (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/surs.h"

namespace surs {

inline char *err_buf() {
    static thread_local char buf[512] = "";
    return buf;
}

inline int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define SURS_HIP_CHECK(expr)                                                                     \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return surs::fail(SURS_E_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
    } while (0)

#define SURS_LAUNCH_CHECK() SURS_HIP_CHECK(hipGetLastError())

#define SURS_REQUIRE(cond, ...)                                   \
    do {                                                          \
        if (!(cond)) return surs::fail(SURS_E_INVALID, __VA_ARGS__); \
    } while (0)

// A value the compiler must materialise as it stands (an empty asm on the register): keeps hipcc from re-forming two instruction
// shapes that were WRONG on MI355X - v_pk_fma_f32 .. op_sel:[0,1,0] (a lost product in a quarter wave with two workgroups per CU,
// NOTES R5.1) and v_fma_mixlo_f16 folds of a product whose fp32 rounding the bit-for-bit tests rely on.  tests/test_isa_pins.py
// disassembles the shipped code object and fails if either shape is back; -DSURS_ABL_NO_ISA_PINS builds the library without the
// pins (only to show that the test then fails).
#ifndef SURS_ABL_NO_ISA_PINS
#define SURS_ISA_PIN(v) asm volatile("" : "+v"(v))
#else
#define SURS_ISA_PIN(v) do { } while (0)
#endif

// Library options (include/surs.h: surs_set_option / surs_get_option): every experiment / A-B switch of the library, in ONE table
// (csrc/surs_api.cpp) instead of getenv calls spread over the kernels' launch code.  The environment variable of an option is
// read once, when the table is first used, as the option's initial value; surs_set_option changes it at any time.
enum Opt {
    OPT_GEMM_X3, OPT_GEMM_BIG, OPT_SPLIT_PARTS, OPT_GEMM_WAVES, OPT_GRID_F32_COLUMNS, OPT_GRID_KERNEL, OPT_GRID_F32_KERNEL, OPT_R_PARTS,
    OPT_GRID_F32_PASSES, OPT_BICUBIC_BLOCK, OPT_MC_EMIT_RECLASSIFY, OPT_POINT_RUNS_SPECULATE, OPT_CONV_TRACE,
    OPT_GEMM_TRACE, OPT_V3_TRACE, OPT_CONV_TALL_MIN_WG, OPT_CONV_WIDE_MIN_WG, OPT_MC_RING, OPT_RVEC_SMALL, OPT_COUNT
};
int option(Opt id);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// "has this kernel's attribute been set on the current device?" - hipFuncSetAttribute belongs to the (function, device)
// pair, so a process-wide flag would leave a second device without its > 64 KB dynamic-LDS limit.  One DeviceOnce per
// kernel instantiation (a function-local static); first() is true exactly once per device.
struct DeviceOnce {
    unsigned long long mask = 0;   // bit d: done on device d (d < 64)
    bool first() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;   // set it again: cheap and harmless
        const unsigned long long bit = 1ull << dev;
        const unsigned long long old = __atomic_fetch_or(&mask, bit, __ATOMIC_ACQ_REL);
        return !(old & bit);
    }
};
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace surs
