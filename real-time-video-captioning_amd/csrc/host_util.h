// Host-side helpers shared by the C-ABI translation units (gitcap.hip, student.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <vector>

struct DevTensor {
    void* p = nullptr;
    std::vector<int64_t> shape;   // logical (unpadded) shape
    bool bf16 = false;            // a GEMM weight (stored as bf16, or as e4m3 + row scales)
    bool loaded = false;
    int64_t bytes = 0;            // device bytes held (incl. row scales)
};

inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }

inline uint16_t host_f2bf(float f) {    // round-to-nearest-even, NaN stays NaN
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// Makes a handle's device current for the duration of an entry point and restores the caller's.
struct DeviceGuard {
    int prev = -1; bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
        if (prev == dev) prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
